"""CPU check of the Winograd F(2x2,3x3) algebra the fused HIP kernel implements (csrc/conv_wino.hip):
the transform matrices, the tile/patch geometry with zero padding and odd sizes, and float32 error of the
transform-domain evaluation against a float64 direct convolution -- next to the float32 direct form's own error."""
import pytest
import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_conv(x, w):
    """x [B,C,H,W], w [O,C,3,3], pad 1, stride 1.  Weights are transformed in float64 and rounded once (as
    wino_weights_kernel does); everything else runs in x.dtype."""
    B, C, H, W = x.shape
    U = (G @ w.double() @ G.T).to(x.dtype)                          # [O,C,4,4]
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    d = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H)).unfold(2, 4, 2).unfold(3, 4, 2)   # [B,C,th,tw,4,4]
    bt, at = BT.to(x.dtype), AT.to(x.dtype)
    V = torch.einsum('ij,bcxyjk,lk->bcxyil', bt, d, bt)
    M = torch.einsum('ocil,bcxyil->boxyil', U, V)
    Y = torch.einsum('ij,boxyjk,lk->boxyil', at, M, at)             # [B,O,th,tw,2,2]
    th, tw = Y.shape[2:4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], th * 2, tw * 2)[:, :, :H, :W]


@pytest.mark.parametrize('shape', [(2, 16, 8, 8), (1, 8, 5, 7), (3, 24, 13, 11), (1, 8, 1, 1), (1, 8, 2, 3)])
def test_winograd_matches_direct_in_float64(shape):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(12, C, 3, 3, generator=g, dtype=torch.float64)
    torch.testing.assert_close(winograd_conv(x, w), F.conv2d(x, w, padding=1), rtol=1e-12, atol=1e-12)


def test_winograd_float32_error_is_that_of_the_direct_form():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 256, 20, 20, generator=g)
    w = torch.randn(64, 256, 3, 3, generator=g) / (256 * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    e_direct = (F.conv2d(x, w, padding=1).double() - ref).pow(2).mean().sqrt().item()
    e_wino = (winograd_conv(x, w).double() - ref).pow(2).mean().sqrt().item()
    assert e_wino < 3.0 * e_direct + 1e-7, (e_wino, e_direct)
    assert (winograd_conv(x, w).double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


# ---- F(4x4,3x3) (csrc/conv_wino4.hip)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def winograd4_conv(x, w):
    B, C, H, W = x.shape
    U = (G4 @ w.double() @ G4.T).to(x.dtype)                         # [O,C,6,6]
    Hp, Wp = (H + 3) // 4 * 4, (W + 3) // 4 * 4
    d = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H)).unfold(2, 6, 4).unfold(3, 6, 4)   # [B,C,th,tw,6,6]
    bt, at = BT4.to(x.dtype), AT4.to(x.dtype)
    V = torch.einsum('ij,bcxyjk,lk->bcxyil', bt, d, bt)
    M = torch.einsum('ocil,bcxyil->boxyil', U, V)
    Y = torch.einsum('ij,boxyjk,lk->boxyil', at, M, at)             # [B,O,th,tw,4,4]
    th, tw = Y.shape[2:4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], th * 4, tw * 4)[:, :, :H, :W]


@pytest.mark.parametrize('shape', [(2, 16, 8, 8), (1, 8, 5, 7), (3, 24, 13, 11), (1, 8, 1, 1), (1, 8, 2, 3)])
def test_winograd4_matches_direct_in_float64(shape):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(12, C, 3, 3, generator=g, dtype=torch.float64)
    torch.testing.assert_close(winograd4_conv(x, w), F.conv2d(x, w, padding=1), rtol=1e-11, atol=1e-11)


def test_winograd4_float32_error_is_bounded():
    """F(4x4,3x3) in float32: ~3e-6 rms / < 6e-5 max on O(1) outputs -- noisier than the direct form (2e-7) because the
    transform constants reach 8, and 30x inside the 1e-4 the detections are held to."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 256, 20, 20, generator=g)
    w = torch.randn(64, 256, 3, 3, generator=g) / (256 * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    err = winograd4_conv(x, w).double() - ref
    assert err.pow(2).mean().sqrt().item() < 6e-6
    assert err.abs().max().item() < 6e-5 * max(1.0, ref.abs().max().item())
