#!/bin/bash
# Round 6 (profiles/r06_experiments.md section 10): N processes x NF back-to-back eval forwards of the fast-profile head (two lanes; encode / condition / ddim_loss calls around the loop): every
# prediction bit-identical to the first?  Arms: the shipped default (the package exports DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 at import: hipGraph replays, runtime fast path off); the variable
# exported as 1 by the caller (the library then enqueues eagerly); graphs forced on WITH the runtime's fast path (the hazard itself).
cd "$(dirname "$0")/../.."
run() { tag=$1; shift; for i in $(seq 1 ${NP:-4}); do env "$@" timeout 400 python tools/gpu/dbg_soak.py 1 ${NF:-1200} 2>&1 | grep "lane_probe=" | cut -c1-420; done | sort | uniq -c | sed "s/^/[$tag] /"; }
NP=6 NF=1200 run shipped_default X=1
NP=4 NF=1200 run fast_path_on_library_goes_eager DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
NP=3 NF=800 run fast_path_on_graphs_forced DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 SOAK_GRAPH=1
