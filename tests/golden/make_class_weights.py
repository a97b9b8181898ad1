"""Golden vectors for calculate_class_weights (reference src/loss.py:163-196): run with the reference tree present.
Inputs (labels / masks of a seeded synthetic dataset) and the reference function's outputs go into class_weights.npz."""
import os

import numpy as np
import torch

from make_golden import HERE, import_reference_loss


def main():
    ref_loss = import_reference_loss()
    rng = np.random.default_rng(17)
    labels = rng.choice(4, size=57, p=[0.5, 0.3, 0.15, 0.05]).astype(np.int64)
    masks = (rng.random((5, 12, 9)) < 0.2).astype(np.int64) + (rng.random((5, 12, 9)) < 0.05).astype(np.int64)
    cls = [{"label": int(v)} for v in labels]
    seg = [{"mask": torch.from_numpy(m)} for m in masks]
    out = {"labels": labels, "masks": masks,
           "cls_w4": ref_loss.calculate_class_weights(cls, 4, "classification").numpy(),
           "cls_w6": ref_loss.calculate_class_weights(cls, 6, "classification").numpy(),   # two absent classes: inf
           "seg_w3": ref_loss.calculate_class_weights(seg, 3, "segmentation").numpy(),
           "seg_w2": ref_loss.calculate_class_weights(seg, 2, "segmentation").numpy()}      # value 2 counts in the total only
    np.savez_compressed(os.path.join(HERE, "class_weights.npz"), **out)
    print({k: v for k, v in out.items() if k.endswith(("w4", "w6", "w3", "w2"))})


if __name__ == "__main__":
    main()
