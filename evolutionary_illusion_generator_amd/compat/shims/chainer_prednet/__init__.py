"""Import shim for the reference's `chainer_prednet` submodule (LanaSina/chainer_prednet): PredNet on the HIP engine."""
