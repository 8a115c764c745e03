"""The reference-side binding really compiles against the reference: integration/check_integration.sh runs the reference's
cmake CONFIGURE step in a scratch directory (to generate config.hpp), patches scratch copies of the five files the patch touches and runs
g++ -fsyntax-only over the reference's own instantiation units that hold them (SortingCountAlgorithm + PartitionsCommand; Bloom / Debloom; MPHF; spans 32 and 64) with
PartitionsByDeviceCommand, and over BloomDevice<LargeInt<1>>, <LargeInt<2>>. Needs /root/reference (build container only): skipped
elsewhere. The link step (patched dbgh5 against libgatbcore.a + libgkc_hip.so) runs when a built reference library is at hand."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GATB_REFERENCE", "/root/reference/gatb-core")
SCRIPT = os.path.join(ROOT, "integration", "check_integration.sh")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "gatb")) or shutil.which("cmake") is None or shutil.which("g++") is None,
                               reason="reference sources / cmake / g++ not available here")


@needs_ref
def test_binding_compiles_against_the_reference_headers(tmp_path_factory):
    scratch = os.environ.get("GKC_INTEGRATION_SCRATCH", "/tmp/gkc_integration")
    out = subprocess.run(["bash", SCRIPT, scratch], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "syntax ok" in out.stdout


@needs_ref
def test_patch_file_is_what_the_generator_produces():
    """integration/gatb-core.device.patch is the diff of the anchored edits (make_patched_sources.py) — and applies to the reference: SortingCountAlgorithm.cpp,
    Bloom.hpp (BloomFactory::createBloom), BloomAlgorithm.cpp, MPHFAlgorithm.cpp, DebloomMinimizerAlgorithm.cpp"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mps", os.path.join(ROOT, "integration", "make_patched_sources.py"))
    mps = importlib.util.module_from_spec(spec); spec.loader.exec_module(mps)
    assert mps.make_diff(REF) == open(os.path.join(ROOT, "integration", mps.PATCH_NAME)).read()
    new = mps.patch(open(os.path.join(REF, mps.REL)).read())
    assert new.count("GATB_WITH_DEVICE_COUNTING") >= 5 and "PartitionsByDeviceCommand<span>" in new
    assert "BloomDevice<T>::usable" in mps.patch_bloom_hpp(open(os.path.join(REF, mps.BLOOM_HPP)).read())
    assert "MphfDevice::build<Type>" in mps.patch_mphf_algo(open(os.path.join(REF, mps.MPHF_ALGO)).read())
    assert "insertSolid" in mps.patch_bloom_algo(open(os.path.join(REF, mps.BLOOM_ALGO)).read())
    assert "contains8Batch" in mps.patch_debloom_algo(open(os.path.join(REF, mps.DEBLOOM_ALGO)).read())


@needs_ref
@pytest.mark.skipif(not os.path.exists(os.environ.get("GATB_BUILD_DIR", "/tmp/gatb_build") + "/lib/Release/libgatbcore.a"),
                    reason="no built reference library (lib/Release/libgatbcore.a) to link against")
def test_patched_dbgh5_links_against_libgkc():
    scratch = os.environ.get("GKC_INTEGRATION_SCRATCH", "/tmp/gkc_integration")
    out = subprocess.run(["bash", SCRIPT, scratch, "--link", os.environ.get("GATB_BUILD_DIR", "/tmp/gatb_build")], capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "link ok" in out.stdout and "U gkc_wait_partition" in out.stdout and "U gkc_push_reads" in out.stdout
    assert "U gkc_bloom_insert_solid" in out.stdout and "U gkc_mphf_build_solid" in out.stdout and "U gkc_bloom_contains8" in out.stdout      # Bloom / MPHF / debloom hunks are in
