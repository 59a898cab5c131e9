"""The C restatement of the native ops (oracle/c/ops_ref.c) against the golden fixtures, for a given build of the library:
imported by tests/test_oracle_golden.py for the plain build, and run as a script (under LD_PRELOAD of the ASAN runtime) for the
-fsanitize=address,undefined build."""
import ctypes
import os
import sys

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_c_oracle(lib_path):
    lib = ctypes.CDLL(lib_path)
    fp = ctypes.POINTER(ctypes.c_float)
    g = np.load(os.path.join(GOLDEN, "ops_upfirdn2d.npz"), allow_pickle=False)
    for name in g["cases"]:
        up, down, p0, p1 = (int(v) for v in g[f"{name}.cfg"])
        x = np.ascontiguousarray(g[f"{name}.x"], dtype=np.float32)
        k = np.ascontiguousarray(g[f"{name}.k"], dtype=np.float32)
        want = g[f"{name}.y"]
        n, c, h, w = x.shape
        y = np.zeros(want.shape, np.float32)
        rc = lib.ref_upfirdn2d(x.ctypes.data_as(fp), k.ctypes.data_as(fp), y.ctypes.data_as(fp), n * c, h, w, 1,
                               k.shape[0], k.shape[1], up, up, down, down, p0, p1, p0, p1)
        assert rc == 0
        np.testing.assert_allclose(y, want, atol=1e-5, err_msg=str(name))
    g = np.load(os.path.join(GOLDEN, "ops_fused_leaky_relu.npz"), allow_pickle=False)
    lib.ref_fused_bias_act.argtypes = [fp, fp, fp, fp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_float, ctypes.c_float]
    for name in g["cases"]:
        x = np.ascontiguousarray(g[f"{name}.x"], dtype=np.float32)
        b = np.ascontiguousarray(g[f"{name}.b"], dtype=np.float32)
        y = np.zeros_like(x)
        step = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
        lib.ref_fused_bias_act(x.ctypes.data_as(fp), b.ctypes.data_as(fp), None, y.ctypes.data_as(fp), x.size, b.size,
                               step, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y, g[f"{name}.y"], atol=1e-6)
    g = np.load(os.path.join(GOLDEN, "postprocess.npz"), allow_pickle=False)
    x = np.ascontiguousarray(g["x"], dtype=np.float32)
    out = np.zeros(g["y"].shape, np.uint8)
    lib.ref_frames_to_u8(x.ctypes.data_as(fp), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 1, 4, 8)
    assert (out == g["y"]).all()


if __name__ == "__main__":
    check_c_oracle(sys.argv[1])
    print("c oracle ok")
