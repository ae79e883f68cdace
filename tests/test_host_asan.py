"""Host-side sanitizer run of the C-ABI's host code (SURVEY.md §5 "ASAN host build of the extension"): csrc/api.hip compiled
as plain C++ with -fsanitize=address,undefined against a mock HIP runtime, launcher stubs that check every range a kernel would read or
write against the live allocations, driven through create -> load -> partial reload -> weights_ready -> reserve -> forwards -> destroy for
all three compute dtypes (tools/host_asan.sh).  CPU only: no GPU, no GPU sanitizer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm's clang++ (host compiler with the sanitizer runtimes) is not installed")
def test_host_code_is_clean_under_asan_ubsan_lsan(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_asan.sh"), str(tmp_path)], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "[host_asan] OK" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail and "LeakSanitizer" not in tail, tail
    for dt in (0, 1, 2):
        assert f"[host_asan] dtype {dt}: clean" in r.stdout
    shutil.rmtree(tmp_path, ignore_errors=True)
