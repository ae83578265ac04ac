"""Shared plumbing of the decode layers: find (or build) the pixel-major head tensor."""
import torch

from ... import ops


def alloc_outputs(nB, n, device):
    return (torch.empty((nB, n, 4), dtype=torch.float32, device=device),
            torch.empty((nB, n), dtype=torch.int64, device=device),
            torch.empty((nB, n), dtype=torch.float32, device=device))


def pack_pixel_major(parts, n_anchor):
    """parts: list of raw tensors [B,A,H,W,c_i] (or [B,H,W,c_i] when n_anchor == 1), any strides.
    Returns a channels-last tensor [B, A*sum(c_i) (padded to 4), H, W] whose channel a*per + c
    follows the order of `parts`, plus (ld, per).  One device copy; used only for raw dicts that
    did not come from this package's heads."""
    if n_anchor == 1 and parts[0].dim() == 4:
        parts = [p.unsqueeze(1) for p in parts]
    cat = torch.cat([p.float() for p in parts], dim=-1)              # [B,A,H,W,per]
    nB, nA, nH, nW, per = cat.shape
    ch = nA * per
    ld = (ch + 3) // 4 * 4
    buf = torch.zeros((nB, nH, nW, ld), dtype=torch.float32, device=cat.device)
    buf[..., :ch] = cat.permute(0, 2, 3, 1, 4).reshape(nB, nH, nW, ch)
    return buf.permute(0, 3, 1, 2)[:, :ch], ld, per
