set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
export NSIDE=4096 LMAX=6144
for rep in 1 2; do
SPIN=2 NCOMP=10 HX_PIPE_ONESET=0 python tools/leg_only.py
SPIN=2 NCOMP=10 HX_PIPE_ONESET=0 HX_LEG_KERNEL=duo python tools/leg_only.py
SPIN=2 NCOMP=10 HX_PIPE_ONESET=0 HX_LEG_KERNEL=duo HX_DUO_PCOL=8 python tools/leg_only.py
SPIN=2 NCOMP=10 HX_PIPE_ONESET=0 HX_LEG_KERNEL=duo HX_DUO_PCOL=16 python tools/leg_only.py
SPIN=2 NCOMP=8 HX_PIPE_ONESET=0 python tools/leg_only.py
SPIN=2 NCOMP=8 HX_PIPE_ONESET=0 HX_LEG_KERNEL=duo python tools/leg_only.py
SPIN=0 NCOMP=10 python tools/leg_only.py
SPIN=0 NCOMP=10 HX_LEG_KERNEL=duo python tools/leg_only.py
SPIN=0 NCOMP=8 python tools/leg_only.py
SPIN=0 NCOMP=8 HX_LEG_KERNEL=duo python tools/leg_only.py
done
) > gpurun_out/r4_t1_duo.log 2>&1
HX_LEG_KERNEL=duo HX_PIPE_ONESET=0 timeout -k 10 900 python -m pytest tests/test_gpu_sht.py -x -q -m gpu > gpurun_out/r4_t1_tests.log 2>&1
tail -5 gpurun_out/r4_t1_tests.log
cat gpurun_out/r4_t1_duo.log | grep -v "^+"
