mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err; echo "bench rc $?"; tail -c 400 gpurun_out/r06/bench_default.json
bash tools/profile_round.sh r06_256 256 voigt
bash tools/profile_round.sh r06_512lam 512 laminate
bash tools/profile_round.sh r06_200 200 voigt
ls gpurun_out/prof_r06_256 gpurun_out/prof_r06_512lam gpurun_out/prof_r06_200
