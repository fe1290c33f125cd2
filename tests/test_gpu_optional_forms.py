"""-m gpu: the opt-in forms of the path's kernels stay correct.  Each switch is read once per process, so a form is exercised by
running a slice of the parity suite in a child process with the switch set:
  DVM_K1_SWEEP=5    pass A of K1 with producer / consumer waves (csrc/dvm_softcorr_sweep2.hip::softcorr_sweep3_kernel)
  DVM_K1_WAVES=4    the second form with 4-wave workgroups, two per compute unit
  DVM_MLP_PERSIST=0 the Deformer MLP's one-workgroup-per-block kernel (the default walks blocks fed by LDS-DMA)
  DVM_MLP_BPW=0     ... and the default kernel with one workgroup per compute unit walking its whole share
(reference: models/loss.py:110-114, 1339-1347; models/model.py:433-452)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

K1 = "test_softcorr_vs_oracle_and_golden or test_softcorr_ragged_shapes or test_softcorr_routed_sweeps_vs_oracle or " \
     "test_argmin_screened_equals_full_scan or test_pair_forward_equals_two_directions or test_softcorr_duplicate_rows"
MLP = "test_deformer_mlp or test_pair_forward_equals_two_directions or test_pair_forward_range_fallback"


@pytest.mark.parametrize("env,select", [({"DVM_K1_SWEEP": "5"}, K1), ({"DVM_K1_WAVES": "4"}, K1),
                                        ({"DVM_MLP_PERSIST": "0"}, MLP), ({"DVM_MLP_BPW": "0"}, MLP)],
                         ids=["k1_split_roles", "k1_four_waves", "mlp_block_kernel", "mlp_one_workgroup_per_cu"])
def test_parity_slice_under_switch(env, select):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                        "-k", select, "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
