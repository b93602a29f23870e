cd /root/repo
X="--steps 20 --warmup 3 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-streams-extra --no-abs-extra"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print('$1', o['value'], o['spread']['step_ms_median'], o['spread']['timed_regions_maps_per_s'], 'b1', (o.get('latency_b1') or {}).get('ms_per_map'))
"; }
for rep in 1 2; do
timeout 300 python bench.py $X 2>/dev/null | pick default
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python bench.py $X 2>/dev/null | pick capture_off
timeout 300 python bench.py $X --no-graph 2>/dev/null | pick no_graph
done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python bench.py $X --variant swin --steps 5 2>/dev/null | pick swin_capture_off
timeout 300 python bench.py $X --variant swin --steps 5 2>/dev/null | pick swin_default
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python bench.py $X --size nyu 2>/dev/null | pick nyu_capture_off
timeout 300 python bench.py $X --size nyu 2>/dev/null | pick nyu_default
