#!/usr/bin/env python3
"""Cuts a list of rocprofv3 counters into passes the hardware can collect at once (a block has a fixed number of counter slots; asking for more makes
rocprofv3 abort at the first HIP call with "error code 38: Request exceeds the capabilities of the hardware to collect" — gpurun_out/pmc_c.log of round 3).
Slots per block on gfx950 as observed on this pool (passes of this size ran; larger ones aborted): SQ 8, TCP 4, TCC 4, TA 2, TD 2, GRBM 2; derived metrics
(FETCH_SIZE, WRITE_SIZE, ... — anything without a known block prefix) expand to several raw TCC counters: one per pass, next to up to 6 SQ counters.
usage: tools/pmc_split.py "<counters>"  -> one pass per line"""
import sys

LIMIT = {"SQ": 8, "TCP": 4, "TCC": 4, "TA": 2, "TD": 2, "GRBM": 2, "SPI": 2, "CPC": 2, "CPF": 2, "GDS": 2}


def block_of(name):
    head = name.split("_")[0]
    return head if head in LIMIT else "DERIVED"


def split(counters):
    passes = []
    for c in counters:
        b = block_of(c)
        for p in passes:
            used = sum(1 for x in p if block_of(x) == b)
            derived = sum(1 for x in p if block_of(x) == "DERIVED")
            if b == "DERIVED":
                ok = derived == 0 and sum(1 for x in p if block_of(x) == "TCC") == 0 and sum(1 for x in p if block_of(x) == "SQ") <= 6
            elif b == "TCC":
                ok = derived == 0 and used < LIMIT[b]
            elif b == "SQ":
                ok = used < (6 if derived else LIMIT[b])
            else:
                ok = used < LIMIT[b]
            if ok:
                p.append(c)
                break
        else:
            passes.append([c])
    return passes


if __name__ == "__main__":
    for p in split(" ".join(sys.argv[1:]).split()):
        print(" ".join(p))
