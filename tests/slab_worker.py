#!/usr/bin/env python3
"""One rank of a slab-decomposed dam break under torch.distributed (launched by tests via torch.distributed.run).
--lib oracle: CPU ranks backed by the CPU oracle (gloo).  --lib product: HIP ranks (nccl, or gloo staging when several
ranks share one GPU).  Rank 0 gathers the owned pieces and writes global-order arrays to --out."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap  # noqa: E402

_bootstrap.load()
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from chrono_wisc_amd import capi, scenes, slab  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--lib", default="oracle")
ap.add_argument("--label", default="tiny")
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--out", required=True)
ap.add_argument("--same-gpu", action="store_true")
ap.add_argument("--isph", type=int, default=0, help="run the implicit scheme with this Jacobi cap instead of WCSPH RK2")
ap.add_argument("--body", action="store_true", help="cylinder drop (one FSI body shared between the ranks) instead of the dam break")
ap.add_argument("--crm", action="store_true", help="CRM soil column (modified Cam-Clay) instead of the dam break: stress companions in every exchange")
ap.add_argument("--poiseuille", action="store_true", help="the reference's Poiseuille unit-test scene, periodic in x: the ranks form a ring (three or more)")
ap.add_argument("--recut", action="store_true", help="re-balance the slab cuts halfway through (collective SlabRank.recut)")
args = ap.parse_args()

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
device = None
if args.lib == "product":
    dev_index = 0 if args.same_gpu else local
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
dist.init_process_group(args.backend, rank=rank, world_size=world)
api = capi.CApi(os.path.join(ROOT, "oracle", "libsph_oracle.so"), "ref_") if args.lib == "oracle" else capi.CApi()
dev_index = 0 if args.same_gpu else local
fsys = None
if args.body:
    s, fsys = scenes.object_drop(api, initial_spacing=0.05, variable_dt=False, initialize=False, device=dev_index if args.lib == "product" else -1)
    fsys.bodies[0].lin_vel = np.array([-1.5, 0.0, -1.0])
    h = 2e-4
elif args.crm:
    h = 1e-4
    s = scenes.granular_column(api, radius=0.1, height=0.2, d0=0.02, box_factor=6.0, dt=h, initialize=False, device=dev_index if args.lib == "product" else -1,
                               mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})
elif args.poiseuille:
    s = scenes.poiseuille(api, bxDim=0.4, t_start=4.0, initialize=False, device=dev_index if args.lib == "product" else -1)
    h = scenes.POISEUILLE["dt"]
elif args.isph:
    s = scenes.dam_break_implicit(api, args.label, max_num_iters=args.isph, atol=0.03, initialize=False, device=dev_index if args.lib == "product" else -1)
    h = 5e-4 * scenes.DAM_BREAK_CONFIGS[args.label][1] / 0.05
else:
    s = scenes.dam_break_named(args.label, api, initialize=False, device=dev_index if args.lib == "product" else -1)
    h = scenes.DAM_BREAK_CONFIGS[args.label][2]
cuts = s.InitializeSlab(rank, world, body_states=[b.state() for b in fsys.bodies] if fsys else None)
r = slab.SlabRank(s, slab.DistTransport(dist, device), rank, world, cuts)
if fsys:
    fsys.advance_cfd = r.step
for k in range(args.steps):
    if args.recut and k == args.steps // 2:
        r.recut()
    if fsys:
        fsys.DoStepDynamics(h)  # exchange loop of ChFsiSystem; OnExchangeSolidForces all-reduces the wrench table
    else:
        r.step(h)
piece = r.owned()
if args.crm:
    piece["stress"], piece["pc"] = s.solver.stress(), s.solver.consolidation()
if fsys:
    piece["wrench"] = np.concatenate(s.GetBodyWrenches()[0])
    piece["body_pos"] = fsys.bodies[0].pos
gathered = [None] * world
dist.gather_object(piece, gathered if rank == 0 else None, dst=0)
if rank == 0:
    g = slab.gather_global(gathered, s.n_global, extra=("stress", "pc") if args.crm else ())
    if fsys:
        for q in gathered:  # every rank holds the same wrench and the same body
            assert np.array_equal(q["wrench"], gathered[0]["wrench"]) and np.array_equal(q["body_pos"], gathered[0]["body_pos"])
        g["wrench"], g["body_pos"] = gathered[0]["wrench"], gathered[0]["body_pos"]
    np.savez(args.out, **g)
dist.barrier()
dist.destroy_process_group()
