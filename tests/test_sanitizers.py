"""CPU: the engine's host concurrency under ThreadSanitizer ("race detection / sanitizers").

tests/tsan_pipeline.cpp drives csrc/host_pipeline.hpp -- the staging threads, slot ring and slice / conversion events of the
stateless pipeline, and the per-shard thread fan-out -- against a fake asynchronous copy engine with HIP's ordering rules, built
with -fsanitize=thread: a slot reused before its copy left it, a raw-record buffer overwritten before its conversion, or a
slice event that does not cover every copy of the slice is a reported data race and a checksum mismatch.  The ASan / UBSan runs
of the arithmetic libraries are in tools/sanitize_host.sh (profiles/r03_sanitizers.txt); they swap shared objects in place and
are kept out of the suite for that reason."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_staging_pipeline_and_shard_fanout_are_race_free(tmp_path):
    exe = str(tmp_path / "tsan_pipeline")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++20", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "2022-entries_amd", "csrc"),
                        "-o", exe, os.path.join(ROOT, "tests", "tsan_pipeline.cpp")], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in r.stderr.lower():
        pytest.skip("libtsan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr and r.stdout.strip().endswith("OK")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_the_harness_catches_a_missing_wait(tmp_path):
    """The harness is only worth something if it fails when the pipeline is wrong: with the wait for the slot's previous copy
    removed, the run must report mismatching bytes or a data race."""
    src = open(os.path.join(ROOT, "2022-entries_amd", "csrc", "host_pipeline.hpp")).read()
    needle = "if (gen) Api::event_sync(ring_ev[slot]);"
    assert src.count(needle) == 1
    (tmp_path / "host_pipeline.hpp").write_text(src.replace(needle, ""))
    exe = str(tmp_path / "tsan_broken")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++20", "-fsanitize=thread", "-pthread", "-I" + str(tmp_path), "-o", exe,
                        os.path.join(ROOT, "tests", "tsan_pipeline.cpp")], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in r.stderr.lower():
        pytest.skip("libtsan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode != 0
