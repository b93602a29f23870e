"""Round 6 debug: 200 eval forwards of the fast-profile Res head -- which iterations differ from the first, by how much, per image; lane_probe from argv"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_zzzz_gpu_soak as S
probe = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
VARIANT = os.environ.get("SOAK_VARIANT", "res")
head, fp, gt = S._kitti_head(VARIANT, profile="fast")
if os.environ.get("SOAK_BATCH"):
    nb = int(os.environ["SOAK_BATCH"]); fp = [f[:nb].contiguous() for f in fp]; gt = gt[:nb].contiguous()
head = head.eval()
be = head._bound.ensure(fp[0].device, head.scheduler)
be.set_option("lane_probe", probe)
if os.environ.get("SOAK_GRAPH") is not None:
    be.set_option("graph", int(os.environ["SOAK_GRAPH"]))
diffs = torch.zeros((N, gt.shape[0]), device="cuda"); first = None
EV = os.environ.get("SOAK_EVENTS", "")        # "timing": a timing-enabled torch event recorded on the caller's stream in front of every forward; "plain": without timing; "once": ONE timing event before the first forward
evs = []
if EV == "once":
    e_ = torch.cuda.Event(enable_timing=True); e_.record(); evs.append(e_)
with torch.no_grad():
    for it in range(N):
        torch.manual_seed(321)
        if EV in ("timing", "plain"):
            e_ = torch.cuda.Event(enable_timing=(EV == "timing")); e_.record(); evs.append(e_)
        out = head(fp, gt, gt > 0, gt_depth_map=gt)
        if first is None:
            first = out["pred"].clone()
        diffs[it] = (out["pred"] - first).abs().flatten(1).max(1)[0]
d = diffs.cpu().numpy()
bad = [i for i in range(N) if d[i].max() > 0]
print(f"variant={VARIANT} B={gt.shape[0]} events={EV or 'none'} lane_probe={probe}: {len(bad)} of {N} forwards differ from the first; first bad {bad[:5]}, last bad {bad[-5:]}; max diff per image {d.max(0)}; overlap {be.counter('lane_overlap')} retries {be.counter('lane_probe_retries')}; graph_default {be.counter('graph_default')} graph_launches {be.counter('graph_launches')} eager_loops {be.counter('eager_loops')} "
      f"(DEBUG_CLR_GRAPH_PACKET_CAPTURE={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')})", flush=True)
if bad:
    import numpy as np
    vals, counts = np.unique(d[bad].max(1), return_counts=True)
    print("distinct max-diffs:", list(zip(vals[:8].tolist(), counts[:8].tolist())))
