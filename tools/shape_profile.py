#!/usr/bin/env python3
"""Per-shape GEMM/conv time inside one full bench step: runs bench.py with SDE_PROF_SHAPES=1 and prints the kernel table.
Name key: c3 = 3x3 conv, g1 = linear/1x1, gg = GEGLU; B = 8-wave 256x320 kernel, S = 4-wave kernel; R = residual epilogue."""
import json, os, subprocess, sys
env = dict(os.environ, SDE_PROF_SHAPES="1")
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "..", "bench.py"), "--steps", "1", "--no-cpu-baseline"] + sys.argv[1:],
                     env=env, capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout, out.stderr); sys.exit(1)
d = json.loads(line[-1])
print(f"images/s {d['value']}  unet ms/call {d['unet_ms_per_call']}  vae ms {d['vae_ms_per_batch']}")
tot = sum(k["total_ms"] for k in d["kernels"])
for k in d["kernels"]:
    if k["total_ms"] < 0.004 * tot: continue
    print(f"{k['name']:44s} n={k['launches']:5d} total {k['total_ms']:8.2f} ms ({100*k['share']:5.1f}%)  avg {k['avg_us']:8.1f} us  "
          f"{k['tflops'] or 0:7.1f} TF  {k['gbs'] or 0:7.1f} GB/s")
