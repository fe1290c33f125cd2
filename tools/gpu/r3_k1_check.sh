# second sweep form after a change: parity subset, bench line + stamps, alpha schedule on both feature sets
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
FORMS="2" bash tools/gpu/r3_k1_forms.sh 2>&1 | tail -6
python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/bench_alpha_form2.txt
