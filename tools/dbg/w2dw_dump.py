import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes
from maua_stylegan2_amd import _lib
torch.set_grad_enabled(False)
gpu = torch.device("cuda:0")
outs = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    cin, cout, h, w, batch = 32, 32, 32, 64, 2
    r = np.random.default_rng(7)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32)).to(gpu)
    s = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, cin))).astype(np.float32)).to(gpu)
    wgt = torch.from_numpy(r.standard_normal((cout, cin, 3, 3)).astype(np.float32)).to(gpu)
    wq = torch.empty(24 * cin * cout, device=gpu)
    lib.maua_pack_weight_wino2d_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert lib.maua_pack_weight_wino2d_f32(wgt.data_ptr(), wq.data_ptr(), cout, cin, None) == 0
    y = torch.zeros(batch, cout, h, w, device=gpu)
    f = lib.maua_modconv3x3_f32
    P = ctypes.c_void_p
    f.argtypes = [P, P, P, ctypes.c_int, P, P] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, P, ctypes.c_int64, P, P, P, P, ctypes.c_int, P]
    nw = torch.zeros(1, device=gpu)
    res = []
    dbg = torch.zeros(2 * 8 * 256 * 8, device=gpu)
    lib.maua_dbg_set.argtypes = [ctypes.c_void_p]
    assert lib.maua_dbg_set(dbg.data_ptr()) == 0
    finals = []
    for rep in range(3):
        y.zero_()
        rc = f(x.data_ptr(), wq.data_ptr(), s.data_ptr(), cin, None, y.data_ptr(), batch, cin, cout, h, w, 5, 1.0, 1, None, 0, nw.data_ptr(), None, None, None, 0, None)
        torch.cuda.synchronize()
        assert rc == 0
        res.append(dbg.cpu().numpy()[:16384].reshape(8, 256, 2, 4).copy())
        second = dbg.cpu().numpy()[16384:].reshape(8, 256, 2, 4).copy()
        dd = res[-1] != second
        if rep == 0:
            print(path, "loop-end dump vs in-epilogue dump differ:", dd.sum())
            if dd.any():
                idx = np.argwhere(dd)
                print(" per lane%16", np.bincount(idx[:, 1] % 16, minlength=16)); print(" per m", np.bincount(idx[:, 2], minlength=2)); print(" per v", np.bincount(idx[:, 3], minlength=4))
        finals.append(y.cpu().numpy().copy())
    outs[path] = res
    outs[path + ":final"] = finals
    print(path, "self-consistent:", [bool((res[0] == r_).all()) for r_ in res])
paths = [p_ for p_ in outs if not p_.endswith(":final")]
fa, fb = outs[paths[0] + ":final"][0], outs[paths[1] + ":final"][0]
print("final outputs differ in", (np.abs(fa - fb) > 1e-3).sum(), "of", fa.size)
a, b = outs[paths[0]][0], outs[paths[1]][0]
d = a != b
print("differing dump entries", d.sum(), "of", d.size)
if d.any():
    idx = np.argwhere(d)
    print(" per wg", np.bincount(idx[:, 0], minlength=8)); print(" per lane%16", np.bincount(idx[:, 1] % 16, minlength=16))
    print(" per wave", np.bincount(idx[:, 1] // 64, minlength=4)); print(" per m", np.bincount(idx[:, 2], minlength=2)); print(" per v", np.bincount(idx[:, 3], minlength=4))
    print(a[tuple(idx[0])], b[tuple(idx[0])])

# ---- which value did the bad build use for M5 = acc[0][5]?  (fuse_act, no noise / bias / demod: t = gain * raw, out = max(t, 0.2 t))
gain = 1.0 * 1.41421356
inv = lambda o: np.where(o >= 0, o, o / 0.2) / gain
bad = np.argwhere(np.abs(fa - fb) > 1e-3)
dump = outs[paths[0]][0]  # [wg, tid, m, v]
print("bad elements", len(bad))
hits = {"zero": 0, "same_lane_other": 0, "none": 0}
shown = 0
for (b, ch, yy, xx) in bad[:400]:
    delta = inv(fa[b, ch, yy, xx]) - inv(fb[b, ch, yy, xx])   # = M5_used - M5_true  (a = bad build, b = good build)
    m, kq, v = ch // 16, (ch % 16) // 4, ch % 4
    ty, r4 = divmod(yy, 16); wv, r2 = divmod(r4, 4); jy, rr = divmod(r2, 2)
    tx, c4 = divmod(xx, 32); jx, px = divmod(c4, 4)
    lane = 16 * kq + 8 * jy + jx
    tid = 64 * wv + lane
    cands = dump[:, tid, m, v]          # over workgroups
    k = int(np.argmin(np.abs(cands + delta)))   # M5_true = -delta if M5_used = 0
    if abs(cands[k] + delta) < 2e-3 * max(1, abs(delta)):
        hits["zero"] += 1
    else:
        # M5_used = some other dump value of the same thread?
        allv = dump[:, tid].reshape(8, 8)
        found = False
        for wg in range(8):
            true = dump[wg, tid, m, v]
            used = true + delta
            if np.any(np.abs(allv[wg] - used) < 2e-3 * max(1, abs(used))):
                found = True
        hits["same_lane_other" if found else "none"] += 1
        if shown < 5:
            shown += 1; print("  e.g.", (b, ch, yy, xx), "delta", delta, "cands", cands)
print(hits)
