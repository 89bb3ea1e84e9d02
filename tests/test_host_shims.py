"""pl-viwo_amd/host/*.h — the C++ adapters a PL-VIWO maintainer drops in (TrackKLT_HIP : ov_core::TrackBase,
TrackLSD_HIP, StateHelperHIP, UpdaterCameraHIP, PlvContext) — through a compiler.  Eigen, OpenCV and the reference tree are not in this
image, so the pass is `g++ -std=c++14 -fsyntax-only` against declaration-only stand-ins of what the adapters include
(tests/host_stub/: names, member types and signatures as the reference declares them): every call into include/plviwo.h is
type-checked against the real header, every use of a reference member against its declared type.  (The first pass of this test found
three errors in the adapters, unbuilt until round 5: OptionsCamera::wh read as a pair, featinit_options as a value, a missing
<cstdlib>.)  Two behavioural contracts are pinned as text: the interpolation polynomial is not rebuilt under use_imu_res
(StateHelper.cpp:171) and a dense R is whitened, never truncated to its diagonal (UpdaterWheel.cpp:130-134 passes a dense 6 x 6)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pl-viwo_amd", "host")
STUB = os.path.join(ROOT, "tests", "host_stub")


@pytest.mark.parametrize("header", sorted(os.path.basename(p) for p in glob.glob(os.path.join(HOST, "*.h"))))
def test_adapter_type_checks_against_the_stand_ins(header):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-x", "c++", "-I" + STUB,
                        "-I" + os.path.join(ROOT, "include"), "-I" + HOST, os.path.join(HOST, header)], capture_output=True, text=True)
    errors = [l for l in r.stderr.splitlines() if "error" in l or ("warning" in l and "#pragma once in main file" not in l)]
    assert r.returncode == 0 and not errors, "\n".join(errors[:20])


def test_reference_call_site_compiles_with_only_the_class_names_changed():
    """UpdaterCamera.cpp:41,44 (`new TrackKLT(cam_intrinsic_model, n_pts, 0, use_stereo, histogram, fast, grid_x, grid_y,
    min_px_dist)`, `new TrackLSD(cam_intrinsic_model, use_stereo, histogram, trackFEATS)`) against the adapters' constructors."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(STUB, "call_sites", "updater_camera_ctor.cpp")
    text = open(src).read()
    assert "new TrackKLT_HIP(state->cam_intrinsic_model, op->n_pts, 0, op->use_stereo, op->histogram, op->fast, op->grid_x, op->grid_y, op->min_px_dist)" in text
    assert "new TrackLSD_HIP(state->cam_intrinsic_model, op->use_stereo, op->histogram, trackFEATS)" in text
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I" + STUB, "-I" + os.path.join(ROOT, "include"),
                        "-I" + HOST, src], capture_output=True, text=True)
    errors = [l for l in r.stderr.splitlines() if "error" in l or "warning" in l]
    assert r.returncode == 0 and not errors, "\n".join(errors[:20])


def test_ekf_update_adapter_keeps_the_reference_contracts():
    src = open(os.path.join(HOST, "StateHelperHIP.h")).read()
    assert "if (!state->op->use_imu_res) state->build_polynomial_data(false);" in src          # StateHelper.cpp:171
    assert "R.isDiagonal()" in src and "llt.matrixL().solve(H)" in src and "llt.matrixL().solve(res)" in src   # dense R: whitened
    assert "R.diagonal()" in src and src.index("R.isDiagonal()") < src.index("R.diagonal()")    # the diagonal is only read when R is diagonal
