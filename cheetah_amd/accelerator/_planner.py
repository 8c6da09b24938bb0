"""The dispatch table of `Segment.track` for a ParticleBeam (reference: cheetah/accelerator/segment.py:545-574 — maximal runs of
skippable elements are merged and applied once, every other element is tracked on its own; cheetah/utils/cache.py:6-68 for what a
cached map may assume).

The reference has ONE way to take a plan item. This engine has several — a whole stretch of lattice in two launches, runs of
non-linear elements in registers, chains of tile-ordered space-charge kicks, the one-call merged run — each with conditions under
which it applies. They used to be an if-ladder inside `Segment._track_internal`; here they are rows of one table:

    Path(name, wants, launch)
      wants(segment, plan, i, kind, item, incoming, state) -> bool     cheap structural test: may this path take plan[i]?
      launch(segment, plan, i, kind, item, incoming, state) -> (outgoing beam, index of the next plan item) | None
                                                                       None = the path declines after all (settings, dtypes,
                                                                       gradients, ...): the next row is asked

The rows are tried in order; the last two never decline. A new way of taking an item — a new element kind with a fused kernel — is
one row here plus its launcher, not another branch of the walk. `state` carries what outlives one item (the buffer of a running
space-charge chain)."""
from __future__ import annotations

from typing import Callable, NamedTuple

import torch

from .. import _ops
from ..particles.particle_beam import ParticleBeam
from .space_charge_kick import SpaceChargeKick


class Path(NamedTuple):
    name: str
    wants: Callable
    launch: Callable


class WalkState:
    """What a walk over the plan carries from item to item."""

    __slots__ = ("chain", "chain_index")

    def __init__(self):
        self.chain = None      # state buffer of a running chain of tile-ordered SpaceChargeKicks: the rows are then in TILE order
        self.chain_index = 0   # position of the next kick in that chain


# ---- [kick, linear run, kick, ...] on one grid: the tile-ordered chain (chx_sc_kick_sorted) ------------------------------------
def _wants_chain(seg, plan, i, kind, item, incoming, state) -> bool:
    return kind == "element" and isinstance(item, SpaceChargeKick) and (state.chain is not None or seg._chain_starts(plan, i, incoming))


def _launch_chain(seg, plan, i, kind, item, incoming, state):
    # the particle rows are sorted by deposit tile once, every kick of the chain works on the ordered rows and the last one restores
    # the caller's particle order. A link needs the run behind it to be applied INSIDE its own particle pass (persistent device plan,
    # no gradients): only then are the sums the gather pass leaves for the next kick's grid the sums of the rows that kick sees, and
    # only then does nothing on the way attach a graph to the beam. Any other run ends the chain at this kick.
    first = state.chain is None
    if first:
        state.chain = _ops.sc_tile_state(incoming.particles.shape[0], item.grid_shape, incoming.particles.dtype, incoming.particles.device)
        state.chain_index = 0
    run = plan[i + 1][1] if i + 1 < len(plan) and plan[i + 1][0] == "run" else None
    fused = seg._chain_run_plan(run, incoming) if run is not None else None
    last = seg._next_chain_kick(plan, i, item, incoming.particles.dtype) is None or (run is not None and fused is None)
    out, step = seg._chain_kick(item, run, fused, incoming, state.chain, first, last, state.chain_index)
    state.chain_index += 1
    if last:
        seg._chain_report(plan, state.chain)
        state.chain = None
    return out, i + step


# ---- [run | active Cavity | BPM | Aperture | Screen]+ : the stretch call (chx_lattice_track*) --------------------------------------
def _wants_stretch(seg, plan, i, kind, item, incoming, state) -> bool:
    return len(plan) - i >= 2 and (kind == "run" or item._is_cavity or item._is_bpm or item._is_aperture or item._is_screen)


def _launch_stretch(seg, plan, i, kind, item, incoming, state):
    return seg._lattice_stretch(plan, i, incoming)


# ---- a linear run in front of non-linear elements rides in their pass ------------------------------------------------------------------
def _wants_run_ahead(seg, plan, i, kind, item, incoming, state) -> bool:
    return kind == "run" and i + 1 < len(plan) and plan[i + 1][0] == "element" \
        and plan[i + 1][1]._tracking_method in ("second_order", "drift_kick_drift")


def _launch_run_ahead(seg, plan, i, kind, item, incoming, state):
    if plan[i + 1][1]._tracking_method == "second_order":
        return seg._second_order_run(plan, i, incoming)
    return seg._dkd_run(plan, i, incoming)


# ---- a run of skippable elements: one composed map, one particle pass (segment.py:545-574) ---------------------------------------
def _wants_run(seg, plan, i, kind, item, incoming, state) -> bool:
    return kind == "run"


def _launch_run(seg, plan, i, kind, item, incoming, state):
    fast = seg._run_apply_fast(item, incoming)
    if fast is None:
        long_run = None
        if len(item.elements) >= seg._PART_MIN_RUN and not (torch.is_grad_enabled() and incoming.particles.requires_grad):
            long_run = seg._run_map_parts(item, incoming.particles, incoming.energy, incoming.species, incoming.s)
        if long_run is None:
            tm, s_out = seg._run_map(item, incoming.energy, incoming.species), seg._run_s(item, incoming.s)
        else:
            tm, s_out = long_run
        new_particles = _ops.apply_map(incoming.particles, tm)
    else:
        new_particles, s_out = fast
    return ParticleBeam(new_particles, incoming.energy, particle_charges=incoming.particle_charges,
                        survival_probabilities=incoming.survival_probabilities, s=s_out, species=incoming.species), i + 1


# ---- consecutive second-order / drift-kick-drift elements in registers -------------------------------------------------------------------
def _wants_second_order(seg, plan, i, kind, item, incoming, state) -> bool:
    return kind == "element" and item._tracking_method == "second_order"


def _launch_second_order(seg, plan, i, kind, item, incoming, state):
    return seg._second_order_run(plan, i, incoming)


def _wants_dkd(seg, plan, i, kind, item, incoming, state) -> bool:
    return kind == "element" and item._tracking_method == "drift_kick_drift"


def _launch_dkd(seg, plan, i, kind, item, incoming, state):
    # a lattice tracked with the Bmad-X maps: consecutive elements go to the device in ONE call (chx_dkd_chain; the per-element
    # Python path costs ~25 us where the kernels take 12-29)
    return seg._dkd_run(plan, i, incoming)


# ---- [SpaceChargeKick, run]: the run's map is applied inside the kick's particle kernel -------------------------------------------
def _wants_kick_then_run(seg, plan, i, kind, item, incoming, state) -> bool:
    return kind == "element" and isinstance(item, SpaceChargeKick) and i + 1 < len(plan) and plan[i + 1][0] == "run"


def _launch_kick_then_run(seg, plan, i, kind, item, incoming, state):
    fused = seg._kick_then_run(item, plan[i + 1][1], incoming)
    return None if fused is None else (fused, i + 2)


# ---- any other element: its own `track` (element.py:149-157) -------------------------------------------------------------------------
def _wants_element(seg, plan, i, kind, item, incoming, state) -> bool:
    return True


def _launch_element(seg, plan, i, kind, item, incoming, state):
    return item._track_internal(incoming), i + 1


PARTICLE_PATHS = (
    Path("space_charge_chain", _wants_chain, _launch_chain),
    Path("lattice_stretch", _wants_stretch, _launch_stretch),
    Path("run_ahead_of_nonlinear", _wants_run_ahead, _launch_run_ahead),
    Path("merged_run", _wants_run, _launch_run),                      # (never declines)
    Path("second_order_run", _wants_second_order, _launch_second_order),
    Path("drift_kick_drift_run", _wants_dkd, _launch_dkd),
    Path("kick_then_run", _wants_kick_then_run, _launch_kick_then_run),
    Path("element", _wants_element, _launch_element),                # (never declines)
)

#: how often each path took an item since the process started (tests / diagnostics: which path does this lattice take?)
TAKEN = {p.name: 0 for p in PARTICLE_PATHS}


def walk_particles(seg, plan, incoming: ParticleBeam) -> ParticleBeam:
    """`Segment.track` of a ParticleBeam: every plan item through the first path of the table that takes it."""
    state = WalkState()
    i, n = 0, len(plan)
    while i < n:
        kind, item = plan[i]
        for path in PARTICLE_PATHS:
            if path.wants(seg, plan, i, kind, item, incoming, state):
                done = path.launch(seg, plan, i, kind, item, incoming, state)
                if done is not None:
                    TAKEN[path.name] += 1
                    incoming, i = done
                    break
    return incoming
