"""The library's tuning knobs and fallback paths (environment variables read once per process) must not change a single
bit: the global-sort MSM path, other accumulate workgroup / chunk sizes, the quotient VM v1 fallback, no hoisted
columns, other shared-subexpression slot counts, the interpreted quotient, the LDS-tile NTT kernel, and the opening's
generator collapse off / forced at several rounds and tail windows.  Each setting runs tests/helpers/env_case.py in its own process; all digests
(MSM results + the bytes of two real ShotCircuit proofs under fixed seeds) must equal the default's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, "tests", "helpers", "env_case.py")

SETTINGS = [
    {"BZH_MSM_GS": "1"},
    {"BZH_ACC_THREADS": "128"},
    {"BZH_ACC_THREADS": "512", "BZH_ACC_CHUNK": "4096"},
    {"BZH_QUOTIENT_V1": "1"},
    {"BZH_NO_HOIST": "1"},
    {"BZH_VM2_CSE": "0"},                                   # (other slot counts = another program: no builtin kernel, the interpreter runs it)
    {"BZH_VM2_CSE": "2"},
    {"BZH_VM2_GLOBAL": "0"},                                # no values kept across gate groups
    {"BZH_VM2_GLOBAL": "2", "BZH_VM2_CSE": "8"},
    {"BZH_QUOTIENT": "interp"},
    {"BZH_NTT_LDS": "1"},
    {"BZH_ACC_SATURATED": "1"},                             # bucket accumulation in saturated 8 x 32 limbs (default: unsaturated 9 x 29, csrc/fe29.cuh)
    {"BZH_ACC_SATURATED": "1", "BZH_ACC_THREADS": "512", "BZH_ACC_CHUNK": "4096"},
    {"BZH_QUOTIENT_SATURATED": "1"},                        # the builtin quotient kernel in saturated limbs (default: its unsaturated-limb flavour)
    {"BZH_RED_WG_MAX": "0"},                                # latency-mode reductions: one wave per segment everywhere
    {"BZH_RED_WG_MAX": "4096"},                             # ... the workgroup flavour everywhere
    {"BZH_MSM_NO_QUAD": "1"},                               # bucket reductions without the four-lanes-per-addition latency mode
    {"BZH_ACC_NO_XCD_MAP": "1"},                            # accumulate workgroups in plain (vector, chunk) grid order
    {"BZH_NO_COMMIT_SHIFT": "1"},                           # grand products committed without taking their constant stretch out
    {"BZH_IPA_COLLAPSE": "0"},                              # no generator collapse (the default collapses from batch 8 on)
    {"BZH_IPA_COLLAPSE": "1"},                              # forced at round 1, also for the 2-proof batch
    {"BZH_IPA_COLLAPSE": "3", "BZH_IPA_TAIL_C": "7"},
    {"BZH_IPA_COLLAPSE": "6", "BZH_IPA_TAIL_C": "11"},
    {"BZH_IPA_COLLAPSE": "3", "BZH_IPA_TAIL_C": "7", "BZH_ACC_SATURATED": "1"},   # the collapse's table expansion in saturated limbs
    {"BZH_IPA_COLLAPSE": "9", "BZH_IPA_TAIL_C": "5"},      # m = 4 folded generators
]


def _digest(extra):
    env = dict(os.environ)
    for k in ("BZH_MSM_GS", "BZH_ACC_THREADS", "BZH_ACC_CHUNK", "BZH_QUOTIENT_V1", "BZH_NO_HOIST", "BZH_VM2_CSE", "BZH_QUOTIENT", "BZH_NTT_LDS",
              "BZH_IPA_COLLAPSE", "BZH_IPA_TAIL_C", "BZH_NO_COMMIT_SHIFT", "BZH_MSM_NO_QUAD", "BZH_ACC_NO_XCD_MAP", "BZH_ACC_SATURATED", "BZH_QUOTIENT_SATURATED", "BZH_RED_WG_MAX", "BZH_VM2_GLOBAL"):
        env.pop(k, None)
    env.update(extra)
    out = subprocess.run([sys.executable, CASE], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST ")]
    assert len(lines) == 1, out.stdout[-500:]
    return lines[0].split()[1]


@pytest.fixture(scope="module")
def default_digest():
    return _digest({})


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_env_setting_changes_no_bit(default_digest, setting):
    assert _digest(setting) == default_digest
