"""Every environment switch of libmatinv_hip.so that selects a kernel or a code path is exercised here once per setting: one
subprocess per setting (the library reads a switch once per process) running a few parity checks against the CPU oracle on the
sizes the switch affects (tests/_switch_worker.py). A kernel that can only be reached through a switch and that no test runs does
not ship: the list below is `grep getenv cuda-matrix-inversion_amd/csrc` minus the switches other tests set
(MATINV_DEVICES, MATINV_DEBUG_REJECTS, MATINV_BLOCKED_WS_MB, MATINV_DETAILED_LOGGING, MATINV_SKIP_GPU, MATINV_LIB)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_switch_worker.py")

# (environment, checks kind:dtype:n[:batch])
SETTINGS = [
    # Gauss-Jordan policy and the screening pass in front of the natural-order kernel
    ({"MATINV_GJ_POLICY": "pivot"}, ["gj_mixed:f64:64:60", "gj_mixed:f32:48:60", "gj_mixed:f64:100:30", "gj_mixed:f64:160:12"]),
    ({"MATINV_GJ_POLICY": "adaptive"}, ["gj_general:f64:64:60", "gj_mixed:f64:64:60", "gj_general:f32:100:30", "gj_mixed:f64:144:12"]),
    ({"MATINV_TILE_SCREEN": "1"}, ["gj_mixed:f64:64:90", "gj_mixed:f64:50:90", "gj_mixed:f32:64:90", "gj_mixed:f32:40:90", "gj_spd:f64:32:40",
                                   "gj_mixed:f64:100:45", "gj_mixed:f64:128:30", "gj_mixed:f64:160:18", "gj_mixed:f32:96:45", "gj_mixed:f32:200:18"]),
    ({"MATINV_TILE_SCREEN": "0"}, ["gj_mixed:f64:64:90", "gj_general:f64:50:40", "gj_mixed:f32:64:90", "gj_mixed:f64:112:30", "gj_mixed:f32:144:18"]),
    # several small matrices per wavefront
    ({"MATINV_ROWLANE_BLOCKS_PER_CU": "2"}, ["gj_spd:f64:8:500", "gj_general:f64:16:300", "gj_spd:f32:12:300"]),
    ({"MATINV_ROWLANE2": "0"}, ["gj_spd:f64:20:60", "gj_mixed:f64:24:60", "gj_spd:f32:25:60"]),
    ({"MATINV_ROWLANE2": "2"}, ["gj_spd:f64:28:60", "gj_mixed:f64:32:60", "gj_spd:f32:30:60"]),
    ({"MATINV_ROWLANE2_GP": "0"}, ["mean:f64:20:60", "variance:f64:24:60", "mean:f32:18:60"]),
    # one wavefront per matrix
    ({"MATINV_TILE_GRID_MULT": "1"}, ["gj_spd:f64:64:5000", "chol:f64:48:5000", "mean:f64:64:5000"]),
    # fused mean / variance dispatch
    ({"MATINV_GP_ROWLANE": "0"}, ["mean:f64:8:100", "variance:f64:16:100", "mean:f32:12:100"]),
    ({"MATINV_GP_BLOCKED": "0"}, ["mean:f64:136:20", "mean:f32:190:20", "variance:f64:130:20"]),
    # blocked multi-launch paths
    ({"MATINV_BGJ_TWO_LEVEL_MIN": "1000"}, ["gj_general:f64:320:12", "gj_general:f32:400:8"]),
    ({"MATINV_BGJ_TWO_LEVEL_MIN": "130"}, ["gj_general:f64:200:12", "gj_general:f32:260:8"]),
    ({"MATINV_BGJ_NB": "64"}, ["gj_general:f64:320:12", "gj_general:f32:512:6"]),
    ({"MATINV_BGJ_NB": "96"}, ["gj_general:f64:400:8"]),
    ({"MATINV_BGP_PAD": "0"}, ["chol:f64:256:12", "mean:f64:320:12", "chol:f32:512:6"]),
    ({"MATINV_BGP_PAIRS": "0"}, ["chol:f64:320:40", "mean:f32:512:40"]),
    ({"MATINV_BGP_PAIRS": "1"}, ["chol:f64:320:6", "mean:f64:256:6"]),
    # fused pipeline, blocked path: Cholesky form forced on a few items / block-LDL^T form forced on many
    ({"MATINV_BGP_LDL": "0"}, ["mean:f32:512:8", "variance:f64:200:20", "mean:f64:1000:3"]),
    ({"MATINV_BGP_GRAPH": "0"}, ["mean:f32:512:8", "variance:f64:200:20"]),
    ({"MATINV_BGP_LDL": "1"}, ["mean:f32:200:600", "variance:f64:320:300", "mean:f32:190:40", "mean:f64:1024:20"]),
    # host-pointer entry points: chunked upload / compute / download overlap forced on a small batch, and off
    ({"MATINV_HOST_PIPELINE": "1"}, ["host:f64:64:3000", "host:f32:32:5000"]),
    ({"MATINV_HOST_PIPELINE": "0"}, ["host:f64:64:3000"]),
]


def _id(setting):
    env, _ = setting
    return ",".join(f"{k}={v}" for k, v in env.items())


@pytest.mark.gpu
@pytest.mark.parametrize("setting", SETTINGS, ids=_id)
def test_switch_selected_path_matches_oracle(setting):
    env, checks = setting
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, WORKER, *checks], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0 and "switch-worker ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_every_switch_of_the_library_is_listed():
    """grep getenv over the kernel sources: a switch that is neither in SETTINGS nor set by another test file fails here."""
    csrc = os.path.join(ROOT, "cuda-matrix-inversion_amd", "csrc")
    found = set()
    for f in os.listdir(csrc):
        with open(os.path.join(csrc, f)) as fh:
            found |= set(re.findall(r'getenv\("(MATINV_[A-Z0-9_]+)"\)', fh.read()))
    elsewhere = {"MATINV_DEVICES", "MATINV_DEBUG_REJECTS", "MATINV_BLOCKED_WS_MB", "MATINV_DETAILED_LOGGING"}
    here = {k for env, _ in SETTINGS for k in env}
    assert found - here - elsewhere == set(), f"switches without a test: {sorted(found - here - elsewhere)}"
    assert here - found == set(), f"settings for switches the library no longer has: {sorted(here - found)}"
    for name in elsewhere:
        hits = [f for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py") and f != os.path.basename(__file__)
                and name in open(os.path.join(ROOT, "tests", f)).read()]
        assert hits, f"{name} is not set by any other test"
