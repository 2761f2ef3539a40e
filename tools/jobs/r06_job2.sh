mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_parity.py -k "chunked" -x -q > gpurun_out/r06/t2.log 2>&1; tail -2 gpurun_out/r06/t2.log
FG_PAIR_STREAMS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -k "chunked" -x -q > gpurun_out/r06/t2s.log 2>&1; tail -2 gpurun_out/r06/t2s.log
O=gpurun_out/r06/ab_pair_chunk_streams.jsonl; : > $O
for nt in 0,0,0,0 2,1,2,1; do
  FG_PAIR_STREAMS=1 FG_PAIR_NT=$nt timeout 600 python tools/ab.py --n 256 --set pair_chunk=0 --set pair_chunk=8 --set pair_chunk=16 --set pair_chunk=32 --set pair_chunk=64 --set pair_chunk=128 --set pair_chunk=0 >> $O 2>&1
done
for nt in 0,0,0,0 2,1,2,1; do
  FG_PAIR_STREAMS=1 FG_PAIR_NT=$nt timeout 900 python tools/ab.py --n 512 --steps 10 --set pair_chunk=0 --set pair_chunk=2 --set pair_chunk=4 --set pair_chunk=8 --set pair_chunk=16 --set pair_chunk=32 --set pair_chunk=0 >> $O 2>&1
done
