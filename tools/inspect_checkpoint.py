#!/usr/bin/env python3
"""Validate a real FoundDiff checkpoint against the layout this engine expects, WITHOUT a GPU.

    python tools/inspect_checkpoint.py checkpoints/FoundDiff/sample/model-400.pt [--dose-clip Dose-CLIP.pth]
                                       [--dim 64 --dim-mults 1,2,4,8 --num-unet 1] [--json]

What it checks, and why (reference: src/DADiff.py:1626-1669, 588-600; README.md:9):
  * the file is the reference's dict {'step', 'model', 'opt0', 'ema', 'scaler'};
  * 'model' = ResidualDiffusion.state_dict(): every LIVE key of founddiff_amd.arch.da_unet_spec is present with the
    expected shape; dead weight (the unused second CLIP, the DA-CLIP text tower, LPIPS) is listed and ignored; anything
    else is UNEXPECTED (another architecture / other constructor arguments);
  * 'ema' in ema-pytorch 0.0.10's layout (install.yaml:186): online_model.* / ema_model.* / initted / step -- the
    ema_model copy is what sampling uses (src/DADiff.py:1818-1822);
  * the 12 schedule buffers, if present, equal the ones init() re-derives (they are overwritten before sampling);
  * `Dose-CLIP.pth` = CLIPIQA.state_dict(): the live visual tower + heads against the same spec.
Exit code 0 = loadable by Trainer.load unchanged, 1 = not.  The report names the first differing keys of each class."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def diff_state(sd, spec, unet_prefixes=("model.unet0.",), extra_dead=()):
    """Compare a state_dict with {key: shape}.  Returns dict(missing, shape_mismatch, dead, unexpected, ok)."""
    from founddiff_amd import arch
    shapes = {k: tuple(v.shape) if hasattr(v, "shape") else None for k, v in sd.items()}

    def want(k):        # spec entries are shapes, or (shape, dtype-name) for integer buffers (BatchNorm counters)
        e = spec[k]
        return tuple(e[0]) if len(e) == 2 and isinstance(e[1], str) else tuple(e)
    missing = [k for k in spec if k not in shapes]
    mism = [(k, shapes[k], want(k)) for k in spec if k in shapes and shapes[k] != want(k)]
    dead, unexpected = [], []
    for k in shapes:
        if k in spec:
            continue
        if any(arch.is_dead_key(k, p) for p in unet_prefixes) or any(k.startswith(p) for p in extra_dead):
            dead.append(k)
        else:
            unexpected.append(k)
    return dict(missing=missing, shape_mismatch=mism, dead=dead, unexpected=unexpected,
                ok=not missing and not mism and not unexpected)


def inspect_checkpoint(path, dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, input_condition=False, clip=None,
                       unsafe=False):
    """`unsafe=False`: the file is read with torch.load(weights_only=True) -- tensors and plain containers only, no
    arbitrary unpickling of an untrusted file.  A checkpoint that carries other pickled objects (the reference saves its
    GradScaler state and version string, src/DADiff.py:1630-1636) needs `unsafe=True` (--unsafe): full unpickling, only
    for files whose origin is trusted."""
    from founddiff_amd import arch
    from founddiff_amd.DADiff import residual_schedule
    clip = clip or arch.RN50
    data, err = _load(path, unsafe)
    if err:
        return {"file": path, "ok": False, "top_level_keys": None, "error": err}
    rep = {"file": path, "top_level_keys": sorted(data) if isinstance(data, dict) else None}
    if not isinstance(data, dict) or "model" not in data:
        rep["error"] = "not the reference's checkpoint dict: no 'model' entry (src/DADiff.py:1630-1636)"
        rep["ok"] = False
        return rep
    rep["step"] = int(data.get("step", -1))
    spec = {}
    prefixes = []
    for u in range(num_unet):
        p = f"model.unet{u}."
        prefixes.append(p)
        spec.update(arch.da_unet_spec(dim, tuple(dim_mults), prefix=p, clip=clip, input_condition=input_condition))
    sched = residual_schedule(1000)
    model_sd = data["model"]
    sched_present = [k for k in sched if k in model_sd]
    body = {k: v for k, v in model_sd.items() if k not in sched}
    rep["model"] = diff_state(body, spec, prefixes, extra_dead=("perceploss.",))
    # the schedule buffers are re-derived by init() (src/DADiff.py:1033-1118, 1818): report whether the stored ones agree
    # with the constructor's or with init()'s values (both are legitimate contents of a checkpoint)
    ctor, init = residual_schedule(1000, after_init=False), residual_schedule(1000, after_init=True)
    rep["schedule_buffers"] = {"present": len(sched_present), "of": len(sched),
                               "match_ctor": all(torch.allclose(model_sd[k].float(), ctor[k], atol=1e-6) for k in sched_present),
                               "match_init": all(torch.allclose(model_sd[k].float(), init[k], atol=1e-6) for k in sched_present)}
    ema = data.get("ema")
    if ema is None:
        rep["ema"] = {"present": False, "note": "no 'ema' entry: sampling would use 'model' (Trainer.load tolerates it)"}
    else:
        em = {k[len("ema_model."):]: v for k, v in ema.items() if k.startswith("ema_model.")}
        on = [k for k in ema if k.startswith("online_model.")]
        other = [k for k in ema if not k.startswith(("ema_model.", "online_model."))]
        e = diff_state({k: v for k, v in em.items() if k not in sched}, spec, prefixes, extra_dead=("perceploss.",))
        e.update(present=True, n_ema_model=len(em), n_online_model=len(on), other_keys=sorted(other),
                 layout_is_ema_pytorch_0_0_10=set(other) <= {"initted", "step"} and bool(em))
        rep["ema"] = e
    rep["ok"] = bool(rep["model"]["ok"] and (not rep["ema"].get("present") or rep["ema"]["ok"]))
    return rep


def _load(path, unsafe):
    """torch.load with the safe unpickler first; full unpickling only behind `unsafe` (see inspect_checkpoint)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True), None
    except Exception as e:                                   # noqa: BLE001 -- report, do not fall back silently
        if not unsafe:
            return None, (f"torch.load(weights_only=True) refused the file ({type(e).__name__}: {e}); "
                          "re-run with --unsafe if the file's origin is trusted")
        return torch.load(path, map_location="cpu", weights_only=False), None


def inspect_dose_clip(path, clip=None, unsafe=False):
    """Dose-CLIP.pth = CLIPIQA(model_type='clipiqa+').state_dict() (src/DADiff.py:595-596): the live part is
    clip_model.visual.*, head1.*, head2.*; prompt_learner.* and the CLIP text side are dead on the sampling path."""
    from founddiff_amd import arch
    clip = clip or arch.RN50
    sd, err = _load(path, unsafe)
    if err:
        return {"file": path, "ok": False, "error": err}
    if isinstance(sd, dict) and "state_dict" in sd and not any(k.startswith("head1") for k in sd):
        sd = sd["state_dict"]
    full = arch.da_unet_spec(64, (1, 2, 4, 8), prefix="", clip=clip)
    spec = {k[len("dose_encoder."):]: v for k, v in full.items() if k.startswith("dose_encoder.")}
    # is_dead_key works on unet-prefixed names: prefix the keys the way they sit inside the UNet
    rep = diff_state({"dose_encoder." + k: v for k, v in sd.items()}, {"dose_encoder." + k: v for k, v in spec.items()},
                     unet_prefixes=("",))
    rep["file"] = path
    return rep


def _short(rep, n=5):
    out = {}
    for k, v in rep.items():
        if isinstance(v, dict):
            out[k] = _short(v, n)
        elif isinstance(v, list) and len(v) > n:
            out[k] = {"count": len(v), "first": v[:n]}
        else:
            out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint", nargs="?")
    ap.add_argument("--dose-clip")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--dim-mults", default="1,2,4,8")
    ap.add_argument("--num-unet", type=int, default=1)
    ap.add_argument("--input-condition", action="store_true")
    ap.add_argument("--unsafe", action="store_true", help="allow full unpickling (torch.load weights_only=False) when the "
                    "safe loader refuses the file: trusted files only")
    ap.add_argument("--json", action="store_true", help="full report as JSON (default: shortened lists)")
    a = ap.parse_args()
    ok = True
    if a.checkpoint:
        rep = inspect_checkpoint(a.checkpoint, a.dim, tuple(int(m) for m in a.dim_mults.split(",")), a.num_unet, a.input_condition, unsafe=a.unsafe)
        print(json.dumps(rep if a.json else _short(rep), indent=1, default=str))
        ok &= rep["ok"]
    if a.dose_clip:
        rep = inspect_dose_clip(a.dose_clip, unsafe=a.unsafe)
        print(json.dumps(rep if a.json else _short(rep), indent=1, default=str))
        ok &= rep["ok"]
    if not a.checkpoint and not a.dose_clip:
        ap.error("give a checkpoint and / or --dose-clip")
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
