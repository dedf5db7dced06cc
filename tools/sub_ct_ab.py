"""A/B of the multi-pass sub-transform kernels (witness_sub_ct 1 vs 2) at the headline shape: per-kernel ms of one proof."""
import os
import subprocess
import sys

for v in sys.argv[1:] or ["1", "2"]:
    env = dict(os.environ, RS_TUNING="witness_sub_ct=" + v)
    out = subprocess.run([sys.executable, "tools/headline_probe.py", "16", "13", "2"], env=env, capture_output=True, text=True)
    print("witness_sub_ct", v)
    print("\n".join(ln for ln in out.stdout.splitlines() if ln.startswith("proof") or "sub_ntt" in ln or "Error" in ln), out.stderr[-400:] if out.returncode else "")
