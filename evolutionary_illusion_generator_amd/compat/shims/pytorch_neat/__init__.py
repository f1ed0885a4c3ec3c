"""Import shim for the reference's `pytorch_neat` submodule (uber-research/PyTorch-NEAT): CPPN node evaluation on the HIP engine."""
