"""CPU: the host layer's reference-compatible C++ API (dbg_assembly_amd/host) -- codec KATs, the
reference's prime quirk, KmerSet maintenance incl. the in-place enlarge -- against the golden KATs
and the oracle.  No GPU, no compute through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dbg_assembly_amd", "lib")


@pytest.fixture(scope="module")
def driver_output(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "libdbgasm_host.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "dbg_assembly_amd", "host")], check=True)
    exe = str(tmp_path_factory.mktemp("hostapi") / "host_api_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "dbg_assembly_amd", "host"),
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_api_test.cpp"),
                    "-o", exe, "-L" + LIBDIR, "-ldbgasm_host", "-ldbgk", "-Wl,-rpath," + LIBDIR], check=True)
    return subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()


def test_kats_equal_reference(driver_output):
    golden = [l.rstrip("\n") for l in open(os.path.join(ROOT, "tests", "golden", "kat.txt"))]
    ours = [l for l in driver_output if l.split("\t")[0] in
            ("seq2bit", "hash_code", "get_next_kmer_depth", "pow_integer", "is_prime", "find_next_prime")]
    assert ours == golden  # same lines, same order as oracle/ref_driver.cpp prints for the real reference


def test_kmerset_maintenance_equals_oracle(driver_output, oracle):
    rows = {l.split("\t")[0]: l.split("\t")[1:] for l in driver_output}
    L = oracle.lib()
    s = L.orc_kmerset_init(1000, 0.7)
    assert rows["init"] == [str(s.contents.size), str(s.contents.max), "16"]
    x = 88172645463325252
    M = (1 << 64) - 1
    node = np.zeros(1, dtype=oracle.NODE_DTYPE)
    for i in range(650):
        x ^= (x << 13) & M
        x ^= x >> 7
        x ^= (x << 17) & M
        node[0] = (x | 1, i, i * 7)
        L.orc_kmerset_add_node(s, node.ctypes.data)
    L.orc_kmerset_enlarge(s, 1)
    assert rows["enlarge1"] == [str(s.contents.size), str(s.contents.max), str(s.contents.count)]
    L.orc_kmerset_enlarge(s, 3000)
    assert rows["enlarge2"] == [str(s.contents.size), str(s.contents.max), str(s.contents.count)]
    assert rows["lookup"] == ["650", "650", "1"]
    assert rows["delete"] == ["550", "1", "1"]
    size = s.contents.size
    arr = np.ctypeslib.as_array(C.cast(s.contents.array, C.POINTER(C.c_uint8)), shape=(size * 16,)).view(oracle.NODE_DTYPE)
    flags = np.ctypeslib.as_array(C.cast(s.contents.nul_flag, C.POINTER(C.c_uint8)), shape=(size // 8 + 1,))
    chk = 0
    occ = np.unpackbits(flags)[:size]
    for i in np.flatnonzero(occ):
        chk = (chk * 1000003 + (int(i) ^ int(arr["kmer"][i]))) & M
    assert rows["layout"] == [str(chk)]  # slot-for-slot the layout the reference's enlarge produces
    assert rows["revcomp"] == ["TACGTT", "C", "1"]
    L.orc_kmerset_free(s)


def test_bench_starts_its_own_ranks_and_relays_their_status():
    """`python bench.py --gpus N` with no torch.distributed environment starts N ranks itself (a torch.distributed.run child).
    Here, without a GPU, every rank stops with "needs an MI355X": the launcher must pass that failure on and print no result."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is visible: the ranks would run")
    assert "starting 2 ranks" in r.stderr
    assert r.returncode != 0 and r.stdout.strip() == ""
