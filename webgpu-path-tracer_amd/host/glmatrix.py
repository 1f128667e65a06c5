"""gl-matrix 3.x compatible subset in numpy (host-side math of the scene builder).

The reference stores every vector/matrix in a ``Float32Array`` and computes in JS doubles
(gl-matrix via CDN: lib/scene.js:1, lib/transform.js:1, lib/camera.js:1, lib/primitives/quad.js:1,
lib/primitives/triangle.js:3).  To reproduce its buffers byte for byte each function below evaluates
the same expression in float64, in the same order, and rounds once on the store to float32.
"""
import math

import numpy as np

EPSILON = 0.000001


def f32(seq):
    return np.asarray(seq, dtype=np.float64).astype(np.float32)


class vec3:
    @staticmethod
    def create():
        return np.zeros(3, np.float32)

    @staticmethod
    def set(out, x, y, z):
        out[0], out[1], out[2] = x, y, z
        return out

    @staticmethod
    def subtract(out, a, b):
        a, b = [float(v) for v in a], [float(v) for v in b]
        out[0], out[1], out[2] = a[0] - b[0], a[1] - b[1], a[2] - b[2]
        return out

    @staticmethod
    def dot(a, b):
        a, b = [float(v) for v in a], [float(v) for v in b]
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    @staticmethod
    def cross(out, a, b):
        ax, ay, az = [float(v) for v in a]
        bx, by, bz = [float(v) for v in b]
        out[0] = ay * bz - az * by
        out[1] = az * bx - ax * bz
        out[2] = ax * by - ay * bx
        return out

    @staticmethod
    def normalize(out, a):
        x, y, z = [float(v) for v in a]
        ln = x * x + y * y + z * z
        if ln > 0:
            ln = 1 / math.sqrt(ln)
        out[0], out[1], out[2] = x * ln, y * ln, z * ln
        return out

    @staticmethod
    def transformMat4(out, a, m):
        x, y, z = [float(v) for v in a]
        m = [float(v) for v in m]
        w = m[3] * x + m[7] * y + m[11] * z + m[15]
        w = w or 1.0
        r0 = (m[0] * x + m[4] * y + m[8] * z + m[12]) / w
        r1 = (m[1] * x + m[5] * y + m[9] * z + m[13]) / w
        r2 = (m[2] * x + m[6] * y + m[10] * z + m[14]) / w
        out[0], out[1], out[2] = r0, r1, r2
        return out

    @staticmethod
    def rotateY(out, a, b, rad):
        a, b = [float(v) for v in a], [float(v) for v in b]
        p = [a[0] - b[0], a[1] - b[1], a[2] - b[2]]
        r0 = p[2] * math.sin(rad) + p[0] * math.cos(rad)
        r1 = p[1]
        r2 = p[2] * math.cos(rad) - p[0] * math.sin(rad)
        out[0], out[1], out[2] = r0 + b[0], r1 + b[1], r2 + b[2]
        return out


class mat4:
    @staticmethod
    def create():
        return np.eye(4, dtype=np.float32).reshape(16).copy()

    @staticmethod
    def identity(out):
        out[:] = 0
        out[0] = out[5] = out[10] = out[15] = 1
        return out

    @staticmethod
    def multiply(out, a, b):
        A = [float(v) for v in a]
        B = [float(v) for v in b]
        res = [0.0] * 16
        for c in range(4):
            b0, b1, b2, b3 = B[4 * c : 4 * c + 4]
            for r in range(4):
                res[4 * c + r] = b0 * A[r] + b1 * A[4 + r] + b2 * A[8 + r] + b3 * A[12 + r]
        out[:] = res
        return out

    mul = multiply

    @staticmethod
    def invert(out, a):
        (a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33) = [float(v) for v in a]
        b00 = a00 * a11 - a01 * a10
        b01 = a00 * a12 - a02 * a10
        b02 = a00 * a13 - a03 * a10
        b03 = a01 * a12 - a02 * a11
        b04 = a01 * a13 - a03 * a11
        b05 = a02 * a13 - a03 * a12
        b06 = a20 * a31 - a21 * a30
        b07 = a20 * a32 - a22 * a30
        b08 = a20 * a33 - a23 * a30
        b09 = a21 * a32 - a22 * a31
        b10 = a21 * a33 - a23 * a31
        b11 = a22 * a33 - a23 * a32
        det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06
        if not det:
            return None
        det = 1.0 / det
        out[:] = [
            (a11 * b11 - a12 * b10 + a13 * b09) * det,
            (a02 * b10 - a01 * b11 - a03 * b09) * det,
            (a31 * b05 - a32 * b04 + a33 * b03) * det,
            (a22 * b04 - a21 * b05 - a23 * b03) * det,
            (a12 * b08 - a10 * b11 - a13 * b07) * det,
            (a00 * b11 - a02 * b08 + a03 * b07) * det,
            (a32 * b02 - a30 * b05 - a33 * b01) * det,
            (a20 * b05 - a22 * b02 + a23 * b01) * det,
            (a10 * b10 - a11 * b08 + a13 * b06) * det,
            (a01 * b08 - a00 * b10 - a03 * b06) * det,
            (a30 * b04 - a31 * b02 + a33 * b00) * det,
            (a21 * b02 - a20 * b04 - a23 * b00) * det,
            (a11 * b07 - a10 * b09 - a12 * b06) * det,
            (a00 * b09 - a01 * b07 + a02 * b06) * det,
            (a31 * b01 - a30 * b03 - a32 * b00) * det,
            (a20 * b03 - a21 * b01 + a22 * b00) * det,
        ]
        return out

    @staticmethod
    def fromTranslation(out, v):
        mat4.identity(out)
        out[12], out[13], out[14] = v[0], v[1], v[2]
        return out

    @staticmethod
    def fromScaling(out, v):
        mat4.identity(out)
        out[0], out[5], out[10] = v[0], v[1], v[2]
        return out

    @staticmethod
    def fromRotation(out, rad, axis):
        x, y, z = [float(v) for v in axis]
        ln = math.hypot(x, y, z)
        if ln < EPSILON:
            return None
        ln = 1 / ln
        x *= ln
        y *= ln
        z *= ln
        s, c = math.sin(rad), math.cos(rad)
        t = 1 - c
        out[:] = [
            x * x * t + c, y * x * t + z * s, z * x * t - y * s, 0,
            x * y * t - z * s, y * y * t + c, z * y * t + x * s, 0,
            x * z * t + y * s, y * z * t - x * s, z * z * t + c, 0,
            0, 0, 0, 1,
        ]
        return out

    @staticmethod
    def targetTo(out, eye, target, up):
        eyex, eyey, eyez = [float(v) for v in eye]
        upx, upy, upz = [float(v) for v in up]
        z0, z1, z2 = eyex - float(target[0]), eyey - float(target[1]), eyez - float(target[2])
        ln = z0 * z0 + z1 * z1 + z2 * z2
        if ln > 0:
            ln = 1 / math.sqrt(ln)
            z0 *= ln
            z1 *= ln
            z2 *= ln
        x0, x1, x2 = upy * z2 - upz * z1, upz * z0 - upx * z2, upx * z1 - upy * z0
        ln = x0 * x0 + x1 * x1 + x2 * x2
        if ln > 0:
            ln = 1 / math.sqrt(ln)
            x0 *= ln
            x1 *= ln
            x2 *= ln
        out[:] = [
            x0, x1, x2, 0,
            z1 * x2 - z2 * x1, z2 * x0 - z0 * x2, z0 * x1 - z1 * x0, 0,
            z0, z1, z2, 0,
            eyex, eyey, eyez, 1,
        ]
        return out
