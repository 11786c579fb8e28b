"""How many distinct cache lines does one gather instruction of the force kernel touch?  CPU-side analysis of the neighbour lists
(built by the oracle library on the host: test infrastructure, no GPU) for candidate record layouts / lane assignments.

  python tools/line_sharing.py [label] [steps] [wave_stride]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import scenes, capi

label = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 8
api = capi.CApi(os.path.join(ROOT, "oracle", "libsph_oracle.so"), "ref_")
s = scenes.dam_break_named(label, api, check_errors=False)
for _ in range(max(steps, 1)):
    s.DoStepDynamics(s.GetStepSize() if hasattr(s, "GetStepSize") else 1e-5)
offs, idx = s.solver.neighbor_csr()
offs = offs.astype(np.int64)
n = len(offs) - 1
cnt = np.diff(offs)
print("rows", n, "mean entries", cnt.mean(), "max", cnt.max())
nw = n // 64
waves = np.arange(0, nw, stride)
K = int(cnt.max())
tot = {}
def add(name, v): tot[name] = tot.get(name, 0) + int(v)
def distinct(a, valid):
    """a, valid: [W, L, K]; number of distinct values among the valid lanes, per (W, K)"""
    b = np.where(valid, a, np.int64(-1))
    b = np.sort(b, axis=1)
    d = (b[:, 1:, :] != b[:, :-1, :]) & (b[:, 1:, :] >= 0)
    return d.sum(axis=1) + (b[:, 0, :] >= 0)
for c0 in range(0, len(waves), 256):
    ws = waves[c0:c0 + 256]
    rows = (ws[:, None] * 64 + np.arange(64)[None, :])           # [W, 64]
    c = cnt[rows]                                                  # [W, 64]
    k = np.arange(1, K)[None, None, :]                             # list position 0 is the row itself
    valid = k < c[:, :, None]
    j = idx[np.minimum(offs[rows][:, :, None] + k, len(idx) - 1)].astype(np.int64)
    instr = valid.any(axis=1)                                      # [W, K-1]: the wave executes this list position
    add("wave_instr", instr.sum()); add("pairs", valid.sum())
    add("distinct_j", distinct(j, valid).sum())
    for sh, nm in ((2, "16B_rec_64B_line"), (3, "16B_rec_128B_line"), (1, "32B_rec_64B_line"), (2, "32B_rec_128B_line")):
        add(nm, distinct(j >> sh, valid).sum())
    # lane-pair scheme: 32-byte records, even rows then odd rows, two lanes per neighbour
    for sh, nm in ((1, "pair32_64B_line"), (2, "pair32_128B_line")):
        add(nm, distinct((j >> sh)[:, 0::2, :], valid[:, 0::2, :]).sum() + distinct((j >> sh)[:, 1::2, :], valid[:, 1::2, :]).sum())
    # model of the counter: an access = a distinct 64-byte line within one hardware quad (4 adjacent lanes) of one instruction
    def quad_lines(lines, valid):          # lines, valid: [W, 64, K] -> sum over 16 quads of the distinct lines in the quad
        W = lines.shape[0]
        l4 = lines.reshape(W, 16, 4, -1); v4 = valid.reshape(W, 16, 4, -1)
        b = np.sort(np.where(v4, l4, np.int64(-1)), axis=2)
        d = ((b[:, :, 1:, :] != b[:, :, :-1, :]) & (b[:, :, 1:, :] >= 0)).sum(axis=2) + (b[:, :, 0, :] >= 0)
        return d.sum()
    add("quadmodel_16B_rec_now", quad_lines(j >> 2, valid))
    add("quadmodel_16B_rec_now_128B", quad_lines(j >> 3, valid))
    # lane-pair scheme under the quad model: instruction 1 = even rows' entries, instruction 2 = odd rows'; lanes (2i, 2i+1) load the two
    # halves of one 32-byte record
    for par in (0, 1):
        jj = np.repeat(j[:, par::2, :], 2, axis=1); vv = np.repeat(valid[:, par::2, :], 2, axis=1)
        add("quadmodel_pair32", quad_lines(jj >> 1, vv))
        add("quadmodel_pair32_128B", quad_lines(jj >> 2, vv))
    # quad scheme: 4 lanes cover 64 bytes = two 32-byte records (entries k, k+1 of one row when consecutive) -- rows 0::4 etc.
    for q in range(4):
        add("quarter_rows_32B_rec_64B_line", distinct((j >> 1)[:, q::4, :], valid[:, q::4, :]).sum())
wi = tot["wave_instr"]
print("sampled waves", len(waves), "wave list positions", wi, "pairs", tot["pairs"], "= %.1f lanes active per position" % (tot["pairs"] / wi))
for k_, v in tot.items():
    if k_ not in ("wave_instr", "pairs"):
        print("%-32s %.2f per wave list position" % (k_, v / wi))
