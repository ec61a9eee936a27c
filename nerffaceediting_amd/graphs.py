"""hipGraph capture of TriPlaneGenerator.synthesis for fixed shapes.

One synthesis() is ~60 kernel launches (13 modulated convs + 7 ToRGB of the backbone, their affines and
demodulations, plane statistics, 1-3 render launches, 6 SR layers); at batch 1 the Python/ctypes launch path
costs as much as the GPU work.  Capturing the sequence once and replaying it removes the host from the loop
("HIP streams and graphs instead of a tracing compiler").  The reference has no counterpart: its viewer calls
synthesis eagerly per frame (viz/renderer.py:441).

Jitter stays random across replays: the renderer reads its Philox key from a device tensor that
`GraphedSynthesis.__call__` refreshes before every replay.
"""
import torch


class GraphedSynthesis:
    def __init__(self, G, batch, neural_rendering_resolution=None, warmup=2, **synthesis_kwargs):
        """Capture G.synthesis(ws[batch,num_ws,w_dim], c[batch,25], **synthesis_kwargs)."""
        self.G = G
        dev = next(G.parameters()).device
        assert dev.type == "cuda", "graph capture needs the GPU path"
        if neural_rendering_resolution is not None:
            G.neural_rendering_resolution = neural_rendering_resolution
        self.ws = torch.zeros(batch, G.backbone.num_ws, G.w_dim, device=dev)
        self.c = torch.zeros(batch, 25, device=dev)
        self.c[:, [0, 5, 10, 15]] = 1.0
        self.c[:, 11] = 2.7
        self.c[:, [16, 20]] = 4.2647
        self.c[:, [18, 21]] = 0.5
        self.c[:, 24] = 1.0
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.kwargs = synthesis_kwargs
        prev_seed = G.renderer.seed_tensor
        G.renderer.seed_tensor = self.seed
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                     # warm-up: packs weights, sizes workspaces
                for _ in range(warmup):
                    G.synthesis(self.ws, self.c, **synthesis_kwargs)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = G.synthesis(self.ws, self.c, **synthesis_kwargs)
        finally:
            G.renderer.seed_tensor = prev_seed

    def __call__(self, ws, c, seed=None):
        """Replay with new latents / cameras.  Returns the captured output dict (tensors are reused by the next
        replay: clone what must outlive it)."""
        self.ws.copy_(ws)
        self.c.copy_(c)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self.seed.fill_(int(seed))
        self.graph.replay()
        return self.out
