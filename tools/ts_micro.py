"""csrc/tile_sort.hip alone on caller-made spans (gs2m_debug_tile_sort): time per launch for a frame of `tiles` spans of lengths in
[lo, hi), per policy; with a GS2M_TS_CLOCK build (GS2M_LIB=.../libts_clock.so) also where a tile's time goes.
usage: python tools/ts_micro.py tiles lo hi"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import gs2m_native
tiles, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
lengths = rng.integers(lo, hi, tiles)
starts = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint32)
n = int(starts[-1])
raw = np.stack([~starts[:-1], starts[1:]], axis=1).astype(np.uint32)
P = 1 << 19
gid = rng.integers(0, P, n).astype(np.uint32)
val = gid | (rng.integers(1, 16, n).astype(np.uint32) << np.uint32(28))
slot = rng.permutation(n).astype(np.uint32)
e_rec = np.zeros((n, 4), np.uint32)
e_rec[slot, 0] = val; e_rec[slot, 1] = rng.integers(0, 1 << 20, n); e_rec[slot, 2] = rng.integers(0x40000000, 0x41000000, n)
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.uint32).view(np.int32).reshape(-1).copy()).to(dev)
t_raw, t_slot, t_erec, t_wrb = T(raw), T(slot), T(e_rec), T(rng.integers(0, 1 << 24, P // 64))
Z = lambda k: torch.zeros((k,), dtype=torch.int32, device=dev)
o_rg, o_pl, o_tmp, o_ql, o_qr, o_qc = Z(2 * tiles), Z(n), Z(max(n, 16 * tiles)), Z(8 * n), Z(4 * n), Z(4 * tiles)
L = gs2m_native.lib()
def call():
    gs2m_native.check(L.gs2m_debug_tile_sort(tiles, t_raw.data_ptr(), o_rg.data_ptr(), t_slot.data_ptr(), t_erec.data_ptr(), t_wrb.data_ptr(), o_pl.data_ptr(),
                                             o_tmp.data_ptr(), o_ql.data_ptr(), o_qr.data_ptr(), o_qc.data_ptr(), gs2m_native.stream_ptr()), "tile sort")
for pol in (0, 1, 2):
    gs2m_native.set_tile_sort_policy(pol)
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"{tiles} tiles of {lo}..{hi} entries ({n} instances), policy {pol}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch set")
gs2m_native.set_tile_sort_policy(0)
o_tmp.zero_(); call(); torch.cuda.synchronize()
c = o_tmp.cpu().numpy().view(np.uint32)[:16 * tiles].reshape(tiles, 16)
if c[:, 14].any():
    t0 = c[:, 15].astype(np.int64); t0 = (t0 - t0.min()) & 0xFFFFFFFF
    print("clock build: per tile, ticks of 10 ns since the tile's start at: records staged / sorted / ties / payload / lists; and the tile's start since the first")
    for name, sel in (("all", slice(None)), ("n<=512", c[:, 14] <= 512), ("513..1024", (c[:, 14] > 512) & (c[:, 14] <= 1024)), (">1024", c[:, 14] > 1024)):
        cc = c[sel]
        if len(cc):
            print(f"  {name:10s} tiles {len(cc):5d}: medians {np.median(cc[:, :5], axis=0).tolist()}  max {cc[:, :5].max(axis=0).tolist()}  start median {np.median(t0[sel]):.0f} max {t0[sel].max()}")
