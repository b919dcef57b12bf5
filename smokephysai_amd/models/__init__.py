from .smokephys_net import SmokePhysNet, ChaosTransformerLayer
from .chaos_attention import ChaosAttention
from .physics_regularizer import PhysicsRegularizer
from .encoder import HipEncoder
from .graphed import GraphedSmokePhysNet

__all__ = ["SmokePhysNet", "ChaosTransformerLayer", "ChaosAttention", "PhysicsRegularizer", "HipEncoder", "GraphedSmokePhysNet"]
