"""MI355X-native training hot path of jnissin/semantic-segmentation.

The module layout mirrors the reference's `src/` package (models, losses, metrics, layers.pooling,
keras_extensions.extended_optimizers / extended_model, trainers, train) so that the reference's config flow can switch
over by changing its imports.  All arithmetic runs in libssmi.so (hand-written HIP for gfx950); PyTorch-ROCm only
provides device memory, streams and torch.distributed (RCCL).

The directory name contains a hyphen, so import it with
    import importlib; ssa = importlib.import_module("semantic-segmentation_amd")
"""
__all__ = ["engine", "models", "losses", "metrics", "layers", "keras_extensions", "trainers"]
