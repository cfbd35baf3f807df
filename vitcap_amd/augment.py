"""Random parameters of the train-time image transform (host side, no GPU).

The reference's training transform is get_inception_train_transform (src/data_layer/transform.py:52-81):
    RandomResizedCrop(train_crop_size, scale=(input_small_scale or 0.08, 1.0))      # ratio (3/4, 4/3), PIL BILINEAR
    ColorJitter(brightness=0.4, contrast=0.4, saturation=0.4)
    RandomHorizontalFlip()                                                         # p = 0.5
    ToTensor(), Normalize(.5, .5)
Those classes live in torchvision, a third-party dependency that is not under /root/reference and not installed in this
image.  What is restated here is the parameter logic of torchvision 0.7.0, the release paired with the reference's
pinned pytorch==1.6.0 (README:20): `RandomResizedCrop.get_params` (10 attempts of area x U(scale), log-uniform aspect
ratio, integer box; central-crop fallback), `ColorJitter.get_params` (one factor per enabled operation from
U(max(0, 1 - v), 1 + v), operations shuffled), flip with probability p.  torchvision 0.7 draws the crop and jitter
numbers from Python's `random` module and the flip from `torch.rand`; here every draw comes from ONE `random.Random`
instance so a sample's augmentation is a function of (seed, epoch, index) whatever the worker layout.  The random
STREAM therefore differs from torchvision's (parity unpinned: nothing to pin against); the distribution is the same,
and the image arithmetic that consumes these numbers is pinned bit for bit against Pillow (oracle/image_oracle.py,
csrc/preproc.hip)."""
import math
import random

OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION = 0, 1, 2


def random_resized_crop_params(rng, height, width, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
    """-> (top, left, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = rng.uniform(scale[0], scale[1]) * area
        aspect_ratio = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = rng.randint(0, height - h)
            j = rng.randint(0, width - w)
            return i, j, h, w
    in_ratio = float(width) / float(height)            # fallback: central crop at the nearest allowed ratio
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def color_jitter_params(rng, brightness=0.4, contrast=0.4, saturation=0.4):
    """-> [(op, factor), ...] in application order."""
    ops = []
    for op, v in ((OP_BRIGHTNESS, brightness), (OP_CONTRAST, contrast), (OP_SATURATION, saturation)):
        if v:
            ops.append((op, rng.uniform(max(0.0, 1.0 - v), 1.0 + v)))
    rng.shuffle(ops)
    return ops


class TrainAugmentation(object):
    """Draws the parameters for one image; `params(h, w, index, epoch)` is deterministic in (seed, epoch, index)."""

    def __init__(self, seed=0, small_scale=None, brightness=0.4, contrast=0.4, saturation=0.4, flip_p=0.5):
        self.seed = int(seed)
        self.scale = (0.08 if small_scale is None else float(small_scale), 1.0)
        self.jitter = (brightness, contrast, saturation)
        self.flip_p = flip_p

    def params(self, height, width, index, epoch=0):
        rng = random.Random((self.seed * 1000003 + epoch) * 2147483647 + index)
        box = random_resized_crop_params(rng, height, width, self.scale)
        ops = color_jitter_params(rng, *self.jitter)
        flip = rng.random() < self.flip_p
        return {'box': box, 'ops': ops, 'flip': flip}
