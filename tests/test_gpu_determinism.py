"""Repeated-run determinism of the fixed-work Newton–Krylov step in the set-ups that exposed two product defects in round 6
(profiles/r06_a_shared_device_root_cause.md):

  * two ranks as two processes on ONE GPU (callback transport and peer arenas) and one rank beside a second copy of the solver —
    the write-after-read race between the wavefronts of the Gram block's factorisation (ss_factor, nk_sstep.hip) only showed when
    another process's kernels disturbed the lock-step of the four wavefronts: 5 of 5 two-rank runs on the callback transport had a
    spurious breakdown, a wrong residual or a hang before the fix;
  * a context on a NON-DEFAULT (non-blocking) stream — the library's set-up memsets ran on the null stream and raced with the first
    kernels on the context's stream (the spare Jacobian value set was zeroed under the fill kernel).

Every case repeats `steps` fixed-work Newton steps of Bratu 512² from u = 0 `trials` times inside one set of processes
(tools/shared_device_probe.py) and demands ONE value of ‖F‖∞, bit for bit, one all-reduce count, nobody hung."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(*args, timeout=240):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shared_device_probe.py"), "--grid", "512", "--trials", "16", "--steps", "6",
                        "--timeout", "150", "--stall-dump", "60"] + list(args), env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(lines[-1])


def _one_value(out, nranks):
    assert out["hung_ranks"] == [], out
    assert len(out["ranks"]) == nranks
    vals = set()
    for r in out["ranks"]:
        assert r["errors"] == [] and r["trials"] == 16, r
        assert r["distinct"] == 1 and len(r["allreduce_counts"]) == 1, r
        vals.add(r["modal"])
    assert len(vals) == 1, vals   # (the norm is all-reduced: every rank holds the same number)
    return vals.pop()


@pytest.mark.parametrize("transport", ["torch", "peer"])
def test_two_ranks_on_one_gpu_repeat_bit_for_bit(transport):
    """the forms round 5 kept away from ranks that share a device (NK_DEVICE_SHARED=0: sweep B on the matrix cores, the resident
    matrix-powers kernel on ranks where the arenas carry it)"""
    _one_value(_probe("--mode", "ranks", "--transport", transport, "--env", "NK_DEVICE_SHARED=0"), 2)


def test_one_rank_beside_a_second_solver_process_repeats_bit_for_bit():
    alone = _one_value(_probe("--mode", "solo"), 1)
    beside = _one_value(_probe("--mode", "solo", "--competitor", "solver", "--competitor-lead", "10"), 1)
    assert alone == beside


def test_a_context_on_a_non_default_stream_matches_the_default_stream():
    alone = _one_value(_probe("--mode", "solo"), 1)
    threaded = _one_value(_probe("--mode", "threads", "--world", "1"), 1)
    assert alone == threaded
