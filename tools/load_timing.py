import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from __graft_entry__ import load_package
pkg = load_package()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # 0 = AUTO; an explicit layout reports why it failed
t0 = time.time()
image = pkg.synth_index(rows, mean_len=8, split_permille=100, seed=42)
print("synth", round(time.time() - t0, 2), "s", flush=True)
t0 = time.time()
tbl = pkg.ColPml.from_bytes(image, layout=layout)
print("open", round(time.time() - t0, 2), "s", tbl.info().layout, flush=True)
t0 = time.time()
tbl.close()
print("close", round(time.time() - t0, 2), "s")
