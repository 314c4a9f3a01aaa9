cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t10_tests.log 2>&1
tail -3 gpurun_out/r4_t10_tests.log
python bench.py > gpurun_out/r4_t10_bench.json 2> gpurun_out/r4_t10_bench.err
tail -c 6000 gpurun_out/r4_t10_bench.json
NSIDE=4096 LMAX=6144 SPIN=2 NCOMP=12 python tools/leg_only.py 2>&1 | grep -v amdgpu.ids
