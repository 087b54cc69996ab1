"""The two `net.registration` symbols test_rpnet.py imports (test_rpnet.py:32,229-230): image
similarity metrics printed next to the Dice scores.  The registration optimisers of the
reference file (affine / Demons / DEEDS, net/registration.py:100-502) are the data-preparation
step in front of the hot path and stay out of scope."""
import torch


def MSE(y_pred, y_true, mask=None):
    """mean squared error (reference net/registration.py:147-154)."""
    value = torch.mean((y_true - y_pred) ** 2)
    if mask is not None:
        value = torch.masked_select(value, mask).mean()
    return value


def NCC(moving_image_valid, fixed_image_valid, mask=None):
    """negative normalised cross-correlation (reference net/registration.py:157-160)."""
    f = fixed_image_valid - torch.mean(fixed_image_valid)
    m = moving_image_valid - torch.mean(moving_image_valid)
    return -1.0 * torch.sum(f * m) / torch.sqrt(torch.sum(f ** 2) * torch.sum(m ** 2) + 1e-10)
