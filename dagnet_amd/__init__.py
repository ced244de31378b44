"""dagnet_amd -- MI355X-native DAG-Net / A-VRNN / VRNN rollout behind the reference's module API.

    from dagnet_amd import DAGNet, AVRNN, VRNN, GAT, GCN
    from dagnet_amd.adjacency_matrix import compute_adjs_distsim
    from dagnet_amd.utils import relative_to_abs, average_displacement_error
    from dagnet_amd.evaluate import evaluate
    from dagnet_amd import data, loop          # loaders' batch layout; train / validate / test epoch functions

The compute path is libdagnet_hip.so (dagnet_amd/csrc, include/dagnet_hip.h); there is no CPU fallback.
"""
from .avrnn import AVRNN
from .baseline import VRNN
from .dagnet import DAGNet
from .gat import GAT, GraphAttentionLayer
from .gcn import GCN, GraphConvolution

__all__ = ["DAGNet", "AVRNN", "VRNN", "GAT", "GCN", "GraphAttentionLayer", "GraphConvolution"]
__version__ = "0.1.0"
