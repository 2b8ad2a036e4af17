"""HIP-graph replay of an inference forward (small-batch serving).

At batch 256 RecNeXt-M3's forward is bound by the GPU (4.0 ms of kernels against 2.7 ms of host-side launches), and a graph of the same
launches replays in the same time (tools/graph_probe.py).  At small batches the ~350 launches of a step are bound by the host: a graph
captured once per input shape replays them with no Python, no ctypes and no allocator in the loop.  The HIP token mixers are plain kernel
launches on the current stream (recnext_amd/ops.py), so they are captured like any ATen operator; their packs, workspace sizes and
kernel attributes are fixed by the warm-up calls before the capture.

    net = build_inference_model("recnext_m3", "cuda")
    run = GraphedInference(net)
    y = run(x)            # captures on the first call with this shape / dtype, replays afterwards

The output is the graph's own buffer: it is overwritten by the next call with the same shape (clone it to keep it).

Weights.  A captured graph holds raw addresses: of the parameters (the GEMMs, the norms) and of the DERIVED packs the token mixers build from
them during the warm-up (RecConv2d._pack, RecAttn2d._pack, DownsampleDwConv._pack), which are rebuilt -- reallocated -- when a parameter's
version changes.  So every call checks a fingerprint of the module's parameters and buffers (address and in-place version counter of each:
load_state_dict, optimizer steps, fold_output_affine and in-place edits all bump a version) and drops every graph when it has changed; the
next call captures again on the new weights.  Each graph also keeps references to the packs it was captured with, so their memory cannot be
handed to another tensor while the graph lives.  What the fingerprint cannot see is a parameter OBJECT replaced after construction
(``module.to(...)``, ``m.weight = nn.Parameter(...)``): call ``reset()`` after such a change (it re-reads the tensor list).
"""
import torch


class GraphedInference:
    def __init__(self, module, warmup=3):
        self.module = module
        self.warmup = warmup
        self._graphs = {}
        self.reset()

    def reset(self):
        """Forget every captured graph and re-read which tensors make up the module's weights."""
        self._graphs.clear()
        self._tensors = [t for t in list(self.module.parameters()) + list(self.module.buffers()) if t is not None]
        self._fingerprint = self._weights_fingerprint()

    def _weights_fingerprint(self):
        return tuple((t.data_ptr(), t._version) for t in self._tensors)

    def _capture(self, x):
        static_x = x.clone(memory_format=torch.preserve_format)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):                      # packs, GEMM solutions, kernel attributes: all before the capture
                self.module(static_x)
        torch.cuda.current_stream(x.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_y = self.module(static_x)
        # the derived packs whose addresses the graph has baked in: kept alive with it
        packs = [getattr(m, a) for m in self.module.modules() for a in ("_pack", "_pad_w", "_pad_b") if getattr(m, a, None) is not None]
        for m in self.module.modules():                       # the fused channel mixer / stem helpers are not Modules: ask them
            for a in ("_fused_mlp", "_fused_stem"):
                helper = m.__dict__.get(a)
                if helper is not None:
                    packs.extend(helper.packs())
        return graph, static_x, static_y, packs

    def __call__(self, x):
        if not x.is_cuda:
            raise RuntimeError("GraphedInference replays a HIP graph: the input must be on the GPU")
        if self.module.training:
            raise RuntimeError("GraphedInference is for inference: call module.eval() first")
        fp = self._weights_fingerprint()
        if fp != self._fingerprint:                              # the weights changed under the graphs: their packs and results are stale
            self._graphs.clear()
            self._fingerprint = fp
        key = (tuple(x.shape), x.dtype, x.device, x.is_contiguous(memory_format=torch.channels_last))
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(x)
        graph, static_x, static_y, _ = entry
        static_x.copy_(x)
        graph.replay()
        return static_y
