"""HIP-graph replay of SmokePhysNet's eval forward.

The reference calls ``model(x)`` eagerly from Python (benchmark.py:33-51, inference.py); at batch 1-4 the forward is a
few hundred short kernels and the host launch path, not the GPU, sets the per-frame time.  On MI355X the idiomatic fix is
a captured hipGraph: the whole forward (libsmokehip's fused encoder launch included -- it is enqueued on the capture
stream like any other kernel) is recorded once per input shape and replayed with one host call.

The in-forward ``torch.randn`` draws of the chaos attention (chaos_attention.py:50-52) are captured through torch's
graph-safe Philox generator, so every replay draws fresh noise exactly like an eager call; pass ``chaos_noise`` to pin
them instead.
"""
from typing import Dict, Optional, Tuple

import torch

from .._lib import require_cuda


class GraphedSmokePhysNet:
    """``g = GraphedSmokePhysNet(model); out = g(x)`` -- same result dictionary as ``model(x)`` in eval mode.

    One graph per (shape, encoder_dtype, pinned-noise?) key is captured on first use.  The returned tensors are the
    graph's static outputs: they are overwritten by the next replay of the same key (``clone=True`` copies them out).
    """

    def __init__(self, model, warmup: int = 2, clone: bool = False, return_features: bool = False):
        self.model = model.eval()
        self.warmup = int(warmup)
        self.clone = bool(clone)
        self.return_features = bool(return_features)
        self._graphs: Dict[Tuple, Tuple] = {}
        self._weights_seen = None
        self.captures = 0            # how many graphs were recorded so far (tests / harnesses: stable once warm)

    def _capture(self, x: torch.Tensor, chaos_noise: Optional[torch.Tensor], encoder_dtype: Optional[str]):
        require_cuda(x.device, "GraphedSmokePhysNet")
        static_x = x.detach().clone()
        static_noise = None if chaos_noise is None else chaos_noise.detach().clone()
        kwargs = dict(return_features=self.return_features, chaos_noise=static_noise, encoder_dtype=encoder_dtype)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, self.warmup)):       # lazy inits (encoder handle, occupancy query, pos-embed cache)
                self.model(static_x, **kwargs)
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.captures += 1
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = self.model(static_x, **kwargs)
        return graph, static_x, static_noise, static_out

    def __call__(self, x: torch.Tensor, chaos_noise: Optional[torch.Tensor] = None,
                 encoder_dtype: Optional[str] = None) -> dict:
        if self.model.training:
            raise RuntimeError("GraphedSmokePhysNet replays the eval forward; call model.eval() first")
        # the encoder's folded weights, the split linear weights and the resized pos-embedding live outside the parameter tensors the graph reads,
        # so a weight update (load_state_dict, an optimizer step) invalidates what was captured.  The fingerprint covers the source
        # tensors only, so the lazily built mirrors of the first forward do not change it (no re-capture on the second call).
        weights_now = self.model.hip_weights_fingerprint()
        if weights_now != self._weights_seen:
            self._graphs.clear()
            self._weights_seen = weights_now
        key = (tuple(x.shape), x.dtype, x.device.index, encoder_dtype, chaos_noise is not None)
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(x, chaos_noise, encoder_dtype)
        graph, static_x, static_noise, static_out = entry
        static_x.copy_(x)
        if static_noise is not None:
            static_noise.copy_(chaos_noise)
        graph.replay()
        if self.clone:
            return {k: v.clone() for k, v in static_out.items()}
        return static_out

    def reset(self):
        """Drop the captured graphs."""
        self._graphs.clear()
