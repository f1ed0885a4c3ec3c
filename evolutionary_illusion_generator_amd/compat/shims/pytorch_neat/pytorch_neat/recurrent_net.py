"""`from pytorch_neat.pytorch_neat.recurrent_net import RecurrentNet` (generate_illusion.py:17): imported by the
reference, never used by it.  Placeholder so the import line resolves."""


class RecurrentNet:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("RecurrentNet is not part of the illusion fitness path")
