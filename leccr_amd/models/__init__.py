"""Mirror of the reference's models/ package (models/__init__.py:1-3)."""
from .xvlm import XVLMBase, build_mlp, allgather, AllGather  # noqa: F401
from .model_retrieval_caption import RetrievalModel  # noqa: F401
