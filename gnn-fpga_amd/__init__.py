"""MI355X-native SegmentClassifier message-passing hot path (reference gnn/model.py).

Layout: `csrc/` HIP kernels + the C-ABI library (`include/gnn_hip.h`), `_lib.py` the
ctypes binding, `model.py` the drop-in nn.Module tree, `hitgraph.py` the index-form
batch/loader, `synth.py` synthetic inputs, `shard.py` event-batch sharding over ranks.
"""
from .synth import HitGraph  # noqa: F401
from .hitgraph import HitGraphBatch  # noqa: F401
from .batcher import GraphStore, batch_generator, merge_graphs  # noqa: F401,E402
