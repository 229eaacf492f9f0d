import sys, os, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from __graft_entry__ import load_package, load_oracle
import test_gpu_parity as t
pkg, oracle = load_package(), load_oracle()
t0 = time.time()
done = 0
for seed0 in range(1000, 1400, 25):
    t.test_fuzz_random_tables_all_layouts(pkg, oracle, seed0)
    done += 25
    print(f"seeds {seed0}..{seed0 + 24} ok ({done} tables, {time.time() - t0:.0f} s)", flush=True)
print("FUZZ-EXTENDED-OK", done)
