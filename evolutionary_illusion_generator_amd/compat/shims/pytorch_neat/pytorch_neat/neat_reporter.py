"""`from pytorch_neat.pytorch_neat.neat_reporter import LogReporter` (generate_illusion.py:16): imported by the
reference, never used by it.  Placeholder so the import line resolves."""


class LogReporter:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("LogReporter is not part of the illusion fitness path; use neat.StdOutReporter")
