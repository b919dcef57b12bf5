from .smokephys_net import SmokePhysNet, ChaosTransformerLayer
from .chaos_attention import ChaosAttention
from .physics_regularizer import PhysicsRegularizer
from .encoder import HipEncoder
from .encoder3d import HipEncoder3D
from .graphed import GraphedSmokePhysNet

__all__ = ["SmokePhysNet", "ChaosTransformerLayer", "ChaosAttention", "PhysicsRegularizer", "HipEncoder", "HipEncoder3D", "GraphedSmokePhysNet"]
