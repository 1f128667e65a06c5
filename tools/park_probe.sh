for p in 0 8 12 16 24; do echo "== PTMI_TAIL_PARK=$p"; PTMI_TAIL_PARK=$p python tools/tail_probe.py -1 2>&1 | grep limit; done
