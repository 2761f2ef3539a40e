set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -k "chunked or staged" -x -q > gpurun_out/r06/t1.log 2>&1; echo rc $? >> gpurun_out/r06/t1.log; tail -3 gpurun_out/r06/t1.log
O=gpurun_out/r06/ab_pair_chunk.jsonl; : > $O
for nt in 0,0,0,0 2,1,2,1 0,1,2,0; do
  FG_PAIR_NT=$nt timeout 600 python tools/ab.py --n 256 --set pair_chunk=0 --set pair_chunk=16 --set pair_chunk=32 --set pair_chunk=64 --set pair_chunk=86 --set pair_chunk=128 --set pair_chunk=0 >> $O 2>&1
done
for nt in 0,0,0,0 2,1,2,1; do
  FG_PAIR_NT=$nt timeout 900 python tools/ab.py --n 512 --steps 10 --set pair_chunk=0 --set pair_chunk=4 --set pair_chunk=8 --set pair_chunk=16 --set pair_chunk=24 --set pair_chunk=32 --set pair_chunk=0 >> $O 2>&1
done
