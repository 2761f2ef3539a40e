for rep in 1 2; do
for pop in 0 1; do echo "populate=$pop"; FG_XFER_POPULATE=$pop timeout 300 python tools/transfer_probe.py 256 2>&1 | grep staged_copy.:.1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:[round(x,1) for x in d[k]] for k in ('prefaulted_ms','fresh_np_empty_ms','fresh_madv_hugepage_ms')})
"; done; done
uptime
