from .ns_dataset import ResidentNSDataset  # noqa: F401
