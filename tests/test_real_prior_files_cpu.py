"""The loaders of the reference's real prior files (smalify_amd/model_io.py: chumpy-free unpickler, load_pose_prior,
unity_shape_prior, family_shape_prior) against the values the reference itself derives from them, as recorded in
tests/golden/reference_golden.npz by tests/golden/make_golden.py (which imports the reference's Prior / SMALFitter).

Runs only where the reference checkout exists (the build container): nothing under /root/reference travels to the GPU box."""
import os
import pickle

import numpy as np
import pytest

from smalify_amd import model_io, synthetic

REF = "/root/reference"
PRIORS = os.path.join(REF, "data", "priors")
pytestmark = pytest.mark.skipif(not os.path.isdir(PRIORS), reason="reference checkout not present")


def test_walking_pose_prior_file(golden):
    prec, mean, mask = model_io.load_pose_prior(os.path.join(PRIORS, "walking_toy_symmetric_pose_prior_with_cov_35parts.pkl"))
    assert prec.dtype == np.float32 and prec.shape == (105, 105)
    assert np.array_equal(prec, golden["pose_prec"])
    assert np.array_equal(mean, golden["pose_mean"])
    assert np.array_equal(mask, golden["pose_mask"])
    # the LF-normalised copy the reference ships for Windows carries the same numbers
    prec_w, mean_w, mask_w = model_io.load_pose_prior(os.path.join(PRIORS, "walking_toy_symmetric_pose_prior_with_cov_35parts_WIN.pkl"))
    assert np.array_equal(prec_w, prec) and np.array_equal(mean_w, mean) and np.array_equal(mask_w, mask)


def test_unity_shape_prior_file(golden):
    prec, mean = model_io.unity_shape_prior(os.path.join(PRIORS, "unity_betas.npz"))
    assert prec.shape == (26, 26) and mean.shape == (26,)
    assert np.array_equal(prec, golden["unity_prec"])
    assert np.array_equal(mean, golden["unity_mean"])


def test_family_shape_prior_from_the_model_pickle_layout(golden, tmp_path):
    """family_shape_prior on a smal_data pickle written in the reference's layout (cluster_means / cluster_cov) and read back
    through the chumpy-free unpickler: the values the reference's SMALFitter derived for shape family 0"""
    _, data, _ = synthetic.synthetic_smal_dicts(seed=0)
    path = tmp_path / "smal_data.pkl"
    with open(path, "wb") as f:
        pickle.dump(data, f, protocol=2)
    prec, mean = model_io.family_shape_prior(model_io.load_pickle(str(path)), 0)
    assert np.array_equal(prec, golden["fam0_prec"])
    assert np.array_equal(mean, golden["fam0_mean"])


def test_joint_limit_table_matches_the_reference_prior():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_joint_limits", os.path.join(REF, "smal_fitter", "priors", "joint_limits_prior.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lp = mod.LimitPrior()
    lo, hi = model_io.joint_limit_table()
    assert np.allclose(lo[:32].reshape(-1), lp.min_values, atol=1e-7) and np.allclose(hi[:32].reshape(-1), lp.max_values, atol=1e-7)
    assert np.all(np.isinf(lo[32:])) and np.all(np.isinf(hi[32:]))
