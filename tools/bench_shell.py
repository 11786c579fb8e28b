#!/usr/bin/env python3
"""Per-step overhead of the C++ co-simulation shell (host/ChFsiSystem_MI: persistent multibody worker + the four exchange hooks) over bare fluid
steps on the C4 cylinder drop: bench_shell.py [d0=0.0054] [steps=200] [generic]
Writes the scene file from the Python scene builder, then runs host/demo_FSI-SPH_ObjectDrop_MI --steps N --no-checks, which prints the wall time of
a coupled step next to the time its fluid phase took by the fluid system's own timer."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
d0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0054
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
step = 2e-5 * d0 / 0.0054
api = capi.CApi()
s, f = scenes.object_drop(api, initial_spacing=d0, variable_dt=False, step_size=step, initial_height=0.9, initialize=False)
exe = os.path.join(ROOT, "chrono-wisc_amd", "host", "demo_FSI-SPH_ObjectDrop_MI")
with tempfile.TemporaryDirectory() as tmp:
    scene = os.path.join(tmp, "drop.scn")
    scenes.write_object_drop_scene(scene, s, f, d0, step, False)
    del s, f
    cmd = [exe, "--scene", scene, "--steps", str(steps), "--no-checks", "--handoff", "2000"] + (["--generic"] if "generic" in sys.argv[3:] else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    print(r.stdout.strip())
    if r.returncode:
        print(r.stderr[-2000:])
        sys.exit(r.returncode)
