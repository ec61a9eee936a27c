"""Plane optimisation through the differentiable renderer: the loop the reference's editing workflow runs with torch autograd
over `utils.decode` (utils.py:165-199 — geometry editing optimises the normalised planes so that the rendered parsing map
matches an edited one, appearance is carried by the plane statistics).  Here every step is one fused forward render and one
`nfe_render_backward`; the optimiser itself is torch.optim on the plane tensor."""
import torch

from . import ops, utils


def segmentation_loss(image_seg, target_labels):
    """Cross-entropy between the rendered 15-channel parsing logits [N,15,R,R] and integer labels [N,R,R]."""
    return torch.nn.functional.cross_entropy(image_seg, target_labels)


def optimize_planes(G, ws, cam, norm_planes, mean, var, loss_fn, steps=100, lr=0.05, optimize="norm", callback=None, **synthesis_kwargs):
    """Optimise tri-planes against `loss_fn(out)` where `out` is `utils.decode`'s dict (image_raw, image_seg, image_depth, and the
    super-resolved `image`, which carries plane gradients for SuperresolutionHybrid8XDC at neural_rendering_resolution 128).

    norm_planes [N,3,32,H,W] (from utils.normalize_plane), mean / var its statistics.  optimize='norm': the normalised planes
    are the leaf and the appearance planes are re-derived from them every step (geometry editing, appearance kept);
    optimize='stats': mean and var are the leaves (appearance editing, geometry kept).  Returns (norm_planes, mean, var, losses).
    """
    assert optimize in ("norm", "stats")
    norm = norm_planes.detach().clone()
    mean, var = mean.detach().clone(), var.detach().clone()
    leaves = [norm] if optimize == "norm" else [mean, var]
    for t in leaves:
        t.requires_grad_(True)
    opt = torch.optim.Adam(leaves, lr=lr)
    losses = []
    for step in range(steps):
        opt.zero_grad(set_to_none=True)
        denorm = utils.denormalize_plane(norm, mean, var)
        out = utils.decode(G, ws, cam, norm, denorm, **synthesis_kwargs)
        loss = loss_fn(out)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))           # synchronises: the place to learn that this step's render or backward was poisoned
        ops.raise_if_handoff_lost()
        if callback is not None:
            callback(step, out, losses[-1])
    return norm.detach(), mean.detach(), var.detach(), losses
