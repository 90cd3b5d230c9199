"""Test infrastructure only: the CPU oracle for GCD's denoising hot path (see svd_unet_ref.py)."""
