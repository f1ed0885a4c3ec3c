"""`from pytorch_neat.pytorch_neat.multi_env_eval import MultiEnvEvaluator` (generate_illusion.py:15): imported by the
reference, never used by it.  Placeholder so the import line resolves."""


class MultiEnvEvaluator:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MultiEnvEvaluator (gym roll-outs) is not part of the illusion fitness path")
