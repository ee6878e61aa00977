"""The C++ host mirror (whisper_amd/host/whisper_host.hpp: ClientProofTag, general_binary_split_test,
run_vdaf_prepare with the reference's names) through its own C++ test programs."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def _binary(name):
    path = os.path.join(CPP, "bin", name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", CPP, os.path.join("bin", name)], check=True, capture_output=True)
    return path


def test_group_testing_cpp_reference_scenario():
    """common/src/grouptest.rs:373-447 (300 tags, 3 corrupted, 2 deleted), both parties as threads."""
    r = subprocess.run([_binary("test_grouptest")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
def test_run_vdaf_prepare_cpp_two_servers():
    """server-batch-prio3/src/main.rs:65-125 on two GPU contexts: tampered clients rejected by both
    servers, aggregate shares unshard to the honest sum."""
    r = subprocess.run([_binary("test_run_vdaf_prepare")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
def test_server_hh_cpp_two_servers():
    """One server-hh repetition (server-hh/src/main.rs:53-110, 255-282) on two GPU contexts through the C++ host mirror:
    malformed clients rejected by both servers (String tags, SHA-256 batch proofs), sketches merge to the honest votes,
    the heavy string is read back (CountSketch::recover)."""
    r = subprocess.run([_binary("test_server_hh")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
def test_funnel_32_threads_two_contexts_and_combine_cpp():
    """include/whisper_gfx950.h's caller funnel: 32 threads (the reference's 32 tokio tasks, server-batch-prio3/src/main.rs:282-335)
    over two contexts with wh_ctx_lock brackets, each getting its own batch's results; the multi-GPU combine behind the C ABI
    with one rank, without RCCL and through RCCL."""
    r = subprocess.run([_binary("funnel_gpu")], capture_output=True, text=True, timeout=300)
    # RCCL prints its version banner first
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), r.stdout + r.stderr
