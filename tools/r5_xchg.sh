# what the neighbour-to-neighbour exchange costs an iteration at one node per GPU, short of the wire: the emulated rank with
# the p2p path run against itself (pack, grouped ncclSend / ncclRecv of every exported record, unpack, joined by update())
tag=${1:-r5/xchg}; mkdir -p gpurun_out/$tag
for rep in 1 2 3; do for x in "" "--force-exchange"; do
  timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 $x 2>gpurun_out/$tag/err.txt | python3 -c "
import json,sys; j=json.loads(sys.stdin.read())
print('emulated rank 3 of 8 %-18s %.4f ms / iteration   exchange: %s   ready-to-done: %s   bytes per exchange: %s' % ('$x' or '(no exchange)', j['ms_per_step'], j['exchange'], j.get('exchange_us_ready_to_done'), (j.get('ranks') or {}).get('bytes_sent_per_exchange')))"
done; done 2>&1 | tee gpurun_out/$tag/self_exchange.txt
tail -2 gpurun_out/$tag/err.txt
