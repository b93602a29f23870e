"""The REFERENCE's own classes of the hot path, run on the CPU (test infrastructure: bench.py's `cpu_baseline` of kind "reference", tests).

Loads duanyiqun/DiffusionDepth's unmodified `DDIMScheduler`, `ScheduledCNNRefine`, `CNNDDIMPipiline` and `DeepDepthTransformWithUpsampling`
through the import shim tests/golden/ref_import.py -- from /root/reference/src where that tree exists (the build container), else from the
sourceless bytecode oracle/ref_py/build_ref.py compiled from it into the git-ignored oracle/_ref/py/ (the GPU box).  Nothing here restates the
algorithm: `ddim_loop_and_decode` is the reference's `CNNDDIMPipiline.__call__` (src/model/head/ddim_depth_estimate_res.py:248-297, its own
torch.randn draw at :277 returning the caller's x_T) followed by `depth_transform.inv_t` (src/model/ops/depth_transform.py:33-35), exactly the
calls the head makes at ...res.py:124-140.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__ may import this module.
"""
from __future__ import annotations

import os
import sys

import torch

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _shim():
    if _GOLDEN not in sys.path:
        sys.path.insert(0, _GOLDEN)
    import ref_import
    return ref_import


_usable = None


def available() -> bool:
    """True when the reference's classes can really be IMPORTED here -- not merely when a directory exists: the staged bytecode is only good for
    the interpreter version that compiled it (STAGED.json records it; a .pyc magic mismatch would otherwise surface as an ImportError in the
    middle of bench.py's cpu_baseline leg), and an import of the tree itself can fail for its own reasons.  One trial import, cached; any
    failure means "not available" and the callers fall back to the port (kind "port")."""
    global _usable
    if _usable is None:
        try:
            R = _shim()
            ok = bool(R.reference_available())
            if ok and R.reference_kind().startswith("staged"):
                import json
                with open(os.path.join(R.STAGED_SRC, "STAGED.json")) as f:
                    ok = json.load(f).get("python") == list(sys.version_info[:2])
            if ok:
                R.load_reference()
            _usable = ok
        except Exception:  # noqa: BLE001
            _usable = False
    return _usable


def kind() -> str:
    return _shim().reference_kind()


def build(sd: dict, variant: str = "res"):
    """(pipeline, codec) = the reference's CNNDDIMPipiline(ScheduledCNNRefine, DDIMScheduler) and DeepDepthTransformWithUpsampling holding `sd`"""
    R = _shim()
    ref = R.load_reference()
    Model = ref.ScheduledCNNRefine if variant == "res" else ref.ScheduledCNNRefineSwin
    Pipe = ref.CNNDDIMPipiline if variant == "res" else ref.CNNDDIMPipilineSwin
    model = R.load_weights(Model(channels_in=256, channels_noise=16).eval(), sd, "model.")
    codec = R.load_weights(ref.DeepDepthTransformWithUpsampling(hidden=16).eval(), sd, "depth_transform.")
    sched = ref.DDIMScheduler(num_train_timesteps=1000, clip_sample=False)        # as the head builds it (...res.py:39-41)
    return Pipe(model, sched), codec


@torch.no_grad()
def ddim_loop_and_decode(pipe, codec, x_T, cond, T: int):
    """x_0, depth = the head's two calls (...res.py:124-140) with the pipeline's own draw of x_T replaced by the caller's tensor"""
    x_T, cond = torch.as_tensor(x_T), torch.as_tensor(cond)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: x_T.clone()
    try:
        x0 = pipe(batch_size=x_T.shape[0], device=x_T.device, dtype=x_T.dtype, shape=tuple(x_T.shape[-3:]),
                  input_args=(cond, None, None, None), num_inference_steps=T, return_dict=False)[0]
    finally:
        torch.randn = real_randn
    return x0, codec.inv_t(x0)
