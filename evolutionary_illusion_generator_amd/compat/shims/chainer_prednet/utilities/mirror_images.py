"""`from chainer_prednet.utilities.mirror_images import mirror, mirror_multiple, TransformationType`
(generate_illusion.py:3, fitness_calculator.py:5).  The reference imports these names and never calls them; they are
provided as small PIL helpers so the import line resolves."""
import os
from enum import IntEnum


class TransformationType(IntEnum):
    Identity = 0
    MirrorH = 1
    MirrorV = 2
    MirrorAndFlip = 3


def mirror(image_path, output_dir, transformation=TransformationType.MirrorH):
    from PIL import Image, ImageOps
    im = Image.open(image_path)
    t = TransformationType(int(transformation))
    if t in (TransformationType.MirrorH, TransformationType.MirrorAndFlip):
        im = ImageOps.mirror(im)
    if t in (TransformationType.MirrorV, TransformationType.MirrorAndFlip):
        im = ImageOps.flip(im)
    os.makedirs(output_dir, exist_ok=True)
    out = os.path.join(output_dir, os.path.basename(image_path))
    im.save(out)
    return out


def mirror_multiple(input_dir, output_dir, transformation=TransformationType.MirrorH, limit=-1):
    names = sorted(n for n in os.listdir(input_dir) if n.lower().endswith((".png", ".jpg", ".jpeg")))
    if limit >= 0:
        names = names[:limit]
    return [mirror(os.path.join(input_dir, n), output_dir, transformation) for n in names]
