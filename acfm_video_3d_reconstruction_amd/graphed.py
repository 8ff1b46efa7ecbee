"""Whole optimiser steps as one hipGraph.

The multiframe training step (multiframe_step.py) is ~30 raster / loss / solve kernels plus several
hundred tiny elementwise launches of camera and loss glue: eager mode is bound by the host's launch
rate, not by the GPU.  Shapes are static from step to step (B clips x T frames x G hypotheses), so
the forward, the backward and the optimiser update are captured once and replayed; every HIP entry
point of libacfm_hip.so is capture-safe (stream-ordered, no host synchronisation, scratch from the
caching allocator's graph pool).

    runner = GraphedStep(lambda inp: step(inp, inp["delta"], textures=inp["tex"], imgs=inp["imgs"]),
                         optimizer, example_inputs, grad_inputs=("delta", "tex"))
    loss = runner(next_inputs)           # copies the inputs into the static buffers, replays
    runner.grads["delta"]                # gradients of non-parameter inputs (the encoder's outputs)

The optimiser must keep its state on the device (torch.optim.Adam(..., capturable=True)).
"""
import torch

from . import ops


def _flat_tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _flat_tensors(v)
    elif isinstance(x, (list, tuple)):
        for v in x:
            yield from _flat_tensors(v)


class GraphedStep:
    def __init__(self, fn, optimizer, example_inputs, grad_inputs=(), n_warmup=3):
        """fn(inputs: dict) -> loss or (loss, aux).  example_inputs: dict of tensors with the shapes
        of every later call.  grad_inputs: keys whose gradient is wanted (leaf inputs that stand for
        the out-of-scope encoder heads).  Parameters and optimiser state are restored after the
        warm-up iterations, so constructing the runner does not train."""
        self.fn, self.opt = fn, optimizer
        self.grad_inputs = tuple(grad_inputs)
        self.static = {}
        for k, v in example_inputs.items():
            if torch.is_tensor(v):
                self.static[k] = v.detach().clone().requires_grad_(k in self.grad_inputs and v.is_floating_point())
            else:
                self.static[k] = v
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_params = [p.detach().clone() for p in params]
        saved_state = {id(p): {k: v.clone() for k, v in optimizer.state.get(p, {}).items() if torch.is_tensor(v)}
                       for p in params}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n_warmup):
                self._one()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self._zero()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.aux = self._one(zero=False)
            self.grads = {k: self.static[k].grad for k in self.grad_inputs}
        with torch.no_grad():                 # undo the warm-up: same tensors, original values
            for p, s in zip(params, saved_params):
                p.copy_(s)
            for p in params:
                for k, v in optimizer.state.get(p, {}).items():
                    if torch.is_tensor(v):
                        old = saved_state[id(p)].get(k)
                        v.copy_(old) if old is not None else v.zero_()

    def _zero(self):
        self.opt.zero_grad(set_to_none=True)
        for k in self.grad_inputs:
            self.static[k].grad = None

    def _one(self, zero=True):
        if zero:
            self._zero()
        out = self.fn(self.static)
        loss, aux = (out[0], out[1:]) if isinstance(out, (tuple, list)) else (out, ())
        loss.backward()
        self.opt.step()
        return loss.detach(), aux

    def __call__(self, inputs):
        with torch.no_grad():
            for k, v in inputs.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v)
        ops.graph_replay(self.graph)      # (+ invalidate_setups(): the replay rewrites tensors without version bumps)
        return self.loss
