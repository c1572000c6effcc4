from .hashgrid import HashEncoder, hash_encode  # noqa: F401
