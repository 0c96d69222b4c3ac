"""
hipGraph replay of the fused RK4 step (single rank) for meshes small enough that the launches, not the
kernels, bound the step -- below roughly 0.5 M dofs when driven from Python (tools/time_rk4_graph.py:
1.7x at 50 k dofs, 1.4x at 118 k, nothing to gain from 1 M dofs up, where consecutive stream launches
overlap their tails and graph nodes do not).

Every launch of a fused step takes fixed device pointers and constants except the source values g(t),
dg/dt of the boundary-facet terms; with ``fus_facet_terms_dev_*`` those are read from device memory, so
the step is captured once (``torch.cuda.CUDAGraph`` = hipStreamBeginCapture / hipGraphLaunch) and
replayed with one 16-byte-per-stage device copy of the step's source values.

The reference drives every launch from Python (cuda/demo_linear_box.py:487-566: 12 launches + 5 host
syncs per stage); this is its launch-bound regime taken to one graph launch per step.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _lib

C_RUNGE = (0.0, 0.5, 0.5, 1.0)


class StepGraphMixin:
    """Adds ``rk4_graph`` to a solver that provides

    ``_graph_state()``        the tensors a step mutates (saved / restored around the warm-up step)
    ``_graph_step_body(dt)``  the launches of one fused step, source values read from ``self._scal[i]``
    ``_graph_scalars(t)``     ``(s1, s2)`` of a stage evaluated at time ``t``
    ``_graph_enter()`` / ``_graph_exit()``   what ``rk4`` does before / after its step loop
    """

    def _step_graph(self, dt):
        g = self._graphs.get(dt)
        if g is None:
            state = self._graph_state()
            saved = [t.clone() for t in state]
            # every kernel of the step once outside the capture (code objects load on first launch, batch
            # plans are built on first use), on state that is put back afterwards
            self._graph_step_body(dt)
            torch.cuda.synchronize()
            for t, s_ in zip(state, saved):
                t.copy_(s_)
            # the captured nodes hold the RAW device pointers of the batch-plan workspaces the operators looked up:
            # keep those workspaces (and the dofmaps they belong to) alive for as long as the graph lives -- the plan
            # cache is bounded and evicts oldest-first, and an evicted workspace that nothing else references would be
            # freed under the graph's feet (replay does not go through the plan registry)
            from . import operators as ops

            g = torch.cuda.CUDAGraph()
            ops._PLANS.start_recording()
            try:
                with torch.cuda.graph(g):
                    self._graph_step_body(dt)
            finally:
                held = ops._PLANS.stop_recording()
            self._graphs[dt] = g
            self._graph_plans = getattr(self, "_graph_plans", {})
            self._graph_plans[dt] = held
        return g

    def rk4_graph(self, start_time, final_time, dt, max_steps=None):
        """``rk4`` with the full-size steps replayed from ONE captured hipGraph.  Same kernels in the same
        order on the same data as ``rk4``.  One rank, fused path; a last shorter step runs through ``rk4``.
        Returns ``(t, steps)``."""
        if not self.fused or self.halo is not None:
            raise _lib.FusGpuError("rk4_graph: single-rank fused path only")
        t, tf = float(start_time), float(final_time)
        rows = []
        while t < tf and (max_steps is None or len(rows) < max_steps) and min(dt, tf - t) == dt:
            rows.append([self._graph_scalars(t + C_RUNGE[i] * dt if self.source_time == "tn" else t) for i in range(4)])
            t += dt
        if rows:
            if not hasattr(self, "_graphs"):
                self._graphs = {}
                self._scal = torch.zeros((4, 2), dtype=self.tdt, device=self.dev)
            table = torch.from_numpy(np.asarray(rows, dtype=np.float64).astype(self.tdt_np)).to(self.dev)
            graph = self._step_graph(dt)
            self._graph_enter()
            for k in range(len(rows)):
                self._scal.copy_(table[k])
                graph.replay()
            self._graph_exit()
        steps = len(rows)
        if t < tf and (max_steps is None or steps < max_steps):
            t, more = self.rk4(t, tf, dt, None if max_steps is None else max_steps - steps)
            steps += more
        return t, steps
