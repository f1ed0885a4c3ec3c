"""Import shim for the reference's `optical_flow` submodule (LanaSina/Optical_Flow_Analyzer): sparse Lucas-Kanade on the HIP engine."""
