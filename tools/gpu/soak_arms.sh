#!/bin/bash
# Round 6 (profiles/r06_experiments.md section 10): N processes x 300 back-to-back eval forwards of the fast-profile head (two lanes, hipGraph replays): every prediction bit-identical to
# the first?  Arms = environment / option settings; the library's lane-overlap probe is on unless LP=0.
cd "$(dirname "$0")/../.."
run() { tag=$1; shift; for i in $(seq 1 ${NP:-6}); do env "$@" timeout 200 python tools/gpu/dbg_soak.py ${LP:-1} 300 2>&1 | grep "lane_probe=" | cut -c1-230; done | sort | uniq -c | sed "s/^/[$tag] /"; }
NP=12 run final_library X=1
NP=4 LP=0 run probe_off X=1
