"""CPU tier: the product's HIP sources (kernels + host logic + C-ABI) compiled
against the SIMT emulator with ASan/UBSan and checked against the oracle.
See tests/emu/hip/hip_runtime.h -- test instrument, not a product path."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_emulated_kernels_match_oracle_under_asan():
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-C", emu, "libcolbwt_emu.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, os.path.join(emu, "run_emu_checks.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "EMU-ALL-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_emulated_kernels_fuzz_all_layouts():
    """Random tables (alphabet 2..9, very long rows, sub-run splits) x random / walk reads, layouts 1-3."""
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-C", emu, "libcolbwt_emu.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, os.path.join(emu, "fuzz_emu.py"), "0", "5"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "FUZZ-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_emulated_granule_store_bookkeeping():
    """csrc/dev_vmm.h with every array through it (granules of 2 MiB): a few fuzz tables, all layouts,
    against the oracle -- granules recycled from array to array, ranges handed out once, everything
    released at close (the emulator's granules are heap objects under ASan)."""
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-C", emu, "libcolbwt_emu.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", COLBWT_VMM_MIN_MB="0", COLBWT_VMM_GRANULE_MB="2")
    out = subprocess.run([sys.executable, os.path.join(emu, "fuzz_emu.py"), "200", "202"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "FUZZ-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
