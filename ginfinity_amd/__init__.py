"""MI355X-native GINE encoder for RNA secondary-structure graphs.

Drop-in for the public surface of ``ginfinity`` 1.2.1
(reference: src/ginfinity/__init__.py:3-36) with the encode hot path running
in hand-written HIP kernels behind the C ABI of ``include/gfy.h``.
"""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  An encoder uses a
# compute stream, a copy stream and up to six copier streams; when the copy stream shares a
# queue, 15 MB D2H copies take 1.2 ms instead of 0.3 and `encode_graphs` 9-12 ms instead of 5.7
# (profiles/README.md, "D2H").  Only effective before the HIP runtime starts (the first CUDA
# call of the process), and only if the host application has not chosen a value itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .records import InputValidationError, RNA
from .spec import (GRAPH_SHARD_FORMAT, GRAPH_SHARD_FORMAT_VERSION,
                   NODE_ROLE_CONTEXT, NODE_ROLE_CORE, GraphCompatibilityError,
                   GraphSpec, GraphValidationError)
from .graph import Graph, GraphBuilder, GraphShard, partition_records
from .shard_io import graph_metadata_path, load_graph_shard, save_graph_shard
from .table import read_rna_table
from .api import Ginfinity, ModelIntegrityError, default_alignment_parameters

__version__ = "1.2.1+mi355x.1"

__all__ = [
    "Ginfinity", "GRAPH_SHARD_FORMAT", "GRAPH_SHARD_FORMAT_VERSION",
    "NODE_ROLE_CONTEXT", "NODE_ROLE_CORE", "Graph", "GraphBuilder",
    "GraphCompatibilityError", "GraphShard", "GraphSpec",
    "GraphValidationError", "InputValidationError", "ModelIntegrityError",
    "RNA", "default_alignment_parameters", "graph_metadata_path",
    "load_graph_shard", "partition_records", "read_rna_table",
    "save_graph_shard", "__version__",
]
