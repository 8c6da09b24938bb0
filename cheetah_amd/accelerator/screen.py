"""Screen (mirror of cheetah/accelerator/screen.py:63-357).

`reading` is produced by HIP kernels reading x / y straight out of the 7-vector rows:
`chx_hist2d` (torch.histogramdd semantics on `torch.linspace` edges) or `chx_cic_deposit`
(2-D cloud-in-cell written directly in the transposed (H, W) layout). The screen misalignment is
subtracted inside the kernels (screen.py:200-212) — no cloned, shifted copy of the beam is made
unless the caller asks for the read beam.
"""

from __future__ import annotations

import torch

from .. import _ops
from ..particles.parameter_beam import ParameterBeam
from ..particles.particle_beam import ParticleBeam
from .element import Element


class Screen(Element):
    """Diagnostic screen."""

    _chx_kind = _ops.KIND["identity"]
    _is_screen = True

    #: beams of at most this many particles get their cloud-in-cell image deposited by the stretch call that tracks them
    #: (`Segment._lattice_stretch`: no further launch, and `reading` has nothing left to do); larger beams are recorded only and the
    #: image is formed when it is asked for — the deposit's atomics then cost more than the launch they save
    _EAGER_IMAGE_PARTICLES = 262_144

    def __init__(self, resolution=(1024, 1024), pixel_size=None, binning=1, misalignment=None,
                 method="cloud-in-cell", kde_bandwidth=None, is_blocking=False, is_active=False, name=None,
                 sanitize_name=None, metadata=None, device=None, dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        assert isinstance(resolution, (tuple, list)) and len(resolution) == 2, "Invalid resolution. Must be a tuple of 2 integers."
        assert method in ("histogram", "kde", "cloud-in-cell"), \
            f"Invalid method {method}. Must be 'histogram', 'kde', or 'cloud-in-cell'."   # screen.py:84-91
        self.register_buffer_or_parameter(
            "pixel_size", pixel_size if pixel_size is not None else torch.tensor((1e-3, 1e-3), **fk))
        self.register_buffer_or_parameter(
            "misalignment", misalignment if misalignment is not None else torch.tensor((0.0, 0.0), **fk))
        if misalignment is None:
            # known to be (0, 0) for as long as this very tensor is in place and unmodified: get_read_beam() then has
            # nothing to shift (a value test would need a device -> host read)
            self.__dict__["_zero_misalignment"] = (self.misalignment, self.misalignment._version)
        self.register_buffer_or_parameter(
            "kde_bandwidth", kde_bandwidth if kde_bandwidth is not None else self.pixel_size[0].clone().detach())
        self.resolution = tuple(resolution)
        self.binning = binning
        self.method = method
        self.is_blocking = is_blocking
        self.is_active = is_active
        self.__dict__["_read_beam"] = None
        self.__dict__["_cached_reading"] = None

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    @property
    def effective_resolution(self) -> tuple[int, int]:
        return (self.resolution[0] // self.binning, self.resolution[1] // self.binning)

    @property
    def effective_pixel_size(self) -> torch.Tensor:
        return self.pixel_size * self.binning

    def _geometry(self, what: str):
        """`extent` / `pixel_bin_edges` are a dozen tiny tensor ops (~100 us of launches): memoised against
        the pixel-size tensor (identity, version), the resolution and the binning."""
        ps = self.pixel_size
        key = (id(ps), ps._version, self.resolution, self.binning)
        cache = self.__dict__.get("_geom_cache")
        if cache is None or cache["key"] != key:
            cache = {"key": key, "pixel_size_ref": ps}
            self.__dict__["_geom_cache"] = cache
        if what not in cache:
            cache[what] = (self._compute_extent() if what == "extent" else self._compute_edges() if what == "edges"
                           else self._compute_gauss_geom() if what == "gauss_geom" else self._compute_sample_counts())
        return cache[what]

    def _compute_gauss_geom(self) -> torch.Tensor:
        """[left, hstep, bottom, vstep] of the sample grid of a ParameterBeam's image (screen.py:276-287)."""
        ext = self.extent
        return torch.stack([ext[0], self.pixel_size[0] * self.binning, ext[2], self.pixel_size[1] * self.binning])

    def _compute_sample_counts(self) -> tuple[int, int]:
        """Number of density samples per axis of a ParameterBeam image: the reference samples on
        `torch.arange(left, right, step)` (screen.py:283-287), whose length is ceil((right - left) / step) evaluated in double
        from the tensors' values — one more than `effective_resolution` whenever the quotient rounds above the integer (one
        host read per screen geometry, cached with it)."""
        import math

        v = torch.cat([self._compute_extent().detach(), (self.pixel_size * self.binning).detach()]).tolist()
        return (int(math.ceil((v[1] - v[0]) / v[4])), int(math.ceil((v[3] - v[2]) / v[5])))

    def _compute_extent(self) -> torch.Tensor:
        return torch.stack([
            -self.resolution[0] * self.pixel_size[0] / 2, self.resolution[0] * self.pixel_size[0] / 2,
            -self.resolution[1] * self.pixel_size[1] / 2, self.resolution[1] * self.pixel_size[1] / 2,
        ])

    @property
    def extent(self) -> torch.Tensor:
        if self.pixel_size.requires_grad:
            return self._compute_extent()
        return self._geometry("extent")

    @property
    def pixel_bin_edges(self) -> tuple[torch.Tensor, torch.Tensor]:
        if self.pixel_size.requires_grad:
            return self._compute_edges()
        return self._geometry("edges")

    def _compute_edges(self) -> tuple[torch.Tensor, torch.Tensor]:
        fk = {"device": self.pixel_size.device, "dtype": self.pixel_size.dtype}
        return (
            torch.linspace(-self.resolution[0] * self.pixel_size[0] / 2, self.resolution[0] * self.pixel_size[0] / 2,
                           int(self.effective_resolution[0]) + 1, **fk),
            torch.linspace(-self.resolution[1] * self.pixel_size[1] / 2, self.resolution[1] * self.pixel_size[1] / 2,
                           int(self.effective_resolution[1]) + 1, **fk),
        )

    @property
    def pixel_bin_centers(self) -> tuple[torch.Tensor, torch.Tensor]:
        ex, ey = self.pixel_bin_edges
        return ((ex[1:] + ex[:-1]) / 2, (ey[1:] + ey[:-1]) / 2)

    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        from .marker import _unaliased

        return _unaliased(self._track_internal(incoming), incoming)   # screen.py:239 `return incoming.clone()`

    def _image_key(self) -> tuple:
        """What an image deposited ahead of its first reading depends on besides the recorded beam: the reference forms the image
        when it is first asked for (screen.py:241-244), with the screen's settings of THAT moment."""
        b = self._buffers
        ps, mis = b.get("pixel_size"), b.get("misalignment")
        if ps is None or mis is None:
            return ()                                   # (trainable geometry: never deposited ahead)
        return (self.resolution, self.binning, self.method, id(ps), ps._version, id(mis), mis._version, mis.dtype, mis.device)

    def _record_stretch(self, record, n: int, species, image, kind: str = "particles", lead: tuple = ()) -> None:
        """Called by the stretch call of `Segment.track` (`chx_lattice_track_screens` / `chx_parameter_lattice_track_screens`):
        `record` is ONE tensor holding the copy of the beam that reached this screen ([rows | charges | survival | energy | s] of
        n particles, or [mu | cov | energy | s | total charge]), `image` the reading deposited by the same call (or None). The
        beam object is built when somebody asks for it."""
        d = self.__dict__
        d["_incoming"] = None
        d["_record"] = (record, n, species, kind, lead)      # (lead: the vector dims of a vectorised beam — its beams one behind the other)
        d["_read_beam"] = None
        mis = self._buffers["misalignment"]
        d["_placed"] = (mis.dtype, mis.device)
        d["_cached_reading"] = image
        d["_eager"] = None if image is None else (self._image_key(), self._buffers["pixel_size"], mis)

    def _incoming_beam(self):
        """The recorded (unshifted) beam; a stretch call's record becomes a beam object here, on first use."""
        d = self.__dict__
        beam = d.get("_incoming")
        if beam is None:
            rec = d.get("_record")
            if rec is not None:
                t, n, species, kind, lead = rec
                if kind == "particles":
                    m = n * _ops.numel(lead)
                    beam = ParticleBeam(t[:7 * m].view(lead + (n, 7)), t[9 * m], particle_charges=t[7 * m:8 * m].view(lead + (n,)),
                                        survival_probabilities=t[8 * m:9 * m].view(lead + (n,)), s=t[9 * m + 1], species=species)
                elif kind == "particles_grad":
                    # the differentiable stretch (cheetah_amd._chxtorch RunScreenTrack): the rows are an output of the node, y = C x
                    # with x free of gradients — a beam property of them hangs on C (`_LinearSource`)
                    rows, q, w, e, s_at, x, C, origin = t
                    rows._chx_lin = _ops._LinearSource(origin, x, C, (), rows._version)
                    beam = ParticleBeam(rows, e, particle_charges=q, survival_probabilities=w, s=s_at, species=species)
                else:
                    beam = ParameterBeam(t[:7], t[7:56].view(7, 7), t[56], total_charge=t[58], s=t[57], species=species)
                d["_incoming"] = beam
                d["_record"] = None
        return beam

    def _track_internal(self, incoming: ParticleBeam) -> ParticleBeam:
        if self.is_active:
            # a snapshot of the unshifted beam is recorded (screen.py:190: later in-place edits of the incoming or
            # outgoing beam must not change the reading); the misalignment is applied inside the image kernels
            # and lazily in get_read_beam()
            self.__dict__["_incoming"] = incoming._snapshot()
            self.__dict__["_record"] = None
            self.__dict__["_eager"] = None
            self.__dict__["_placed"] = (self.misalignment.dtype, self.misalignment.device)
            self.__dict__["_read_beam"] = None
            self.__dict__["_cached_reading"] = None
        if self.is_active and self.is_blocking:
            if isinstance(incoming, ParameterBeam):
                return ParameterBeam(incoming.mu, incoming.cov, incoming.energy,
                                     total_charge=torch.zeros_like(incoming.total_charge), s=incoming.s,
                                     species=incoming.species)
            return ParticleBeam(incoming.particles, incoming.energy, particle_charges=incoming.particle_charges,
                                survival_probabilities=torch.zeros_like(incoming.survival_probabilities),
                                s=incoming.s, species=incoming.species)
        return incoming._view()

    @property
    def reading(self) -> torch.Tensor:
        """Image of shape (…, height, width)."""
        d = self.__dict__
        eager = d.get("_eager")
        if eager is not None:
            # the image the tracking call deposited: valid if the screen still is what it was then
            d["_eager"] = None
            if eager[0] == self._image_key():
                return d["_cached_reading"]
            d["_cached_reading"] = None
        # Was the screen moved with .to() / .double() / .cuda() since the beam was recorded? Then the recorded beam and a cached
        # image follow it, like the reference's read beam, which is a sub-module of the screen (test_screen.py:137-159). A beam
        # whose dtype merely differs from the screen's is left alone: the reference's image then has the BEAM's dtype.
        now = (self.misalignment.dtype, self.misalignment.device)
        placed = self.__dict__.get("_placed", now)
        moved = placed != now
        if moved:
            self.__dict__["_placed"] = now
        cached = self.__dict__.get("_cached_reading")
        if cached is not None:
            if moved:
                cached = cached.to(device=now[1], dtype=now[0])
                self.__dict__["_cached_reading"] = cached
            return cached
        beam = self._incoming_beam()
        if beam is not None and moved:
            beam = beam.to(device=now[1], dtype=now[0])
            self.__dict__["_incoming"] = beam
            self.__dict__["_read_beam"] = None
        if (beam is not None and not isinstance(beam, ParameterBeam) and self.method == "cloud-in-cell"
                and torch.promote_types(beam.particles.dtype, now[0]) != beam.particles.dtype):
            # cloud_in_cell.py: positions normalised with the screen's (wider) extent are scattered with the beam's (narrower)
            # charges — torch refuses the mixed scatter; the same combination raises here
            raise RuntimeError("scatter(): Expected self.dtype to be equal to src.dtype "
                               f"(screen {now[0]}, beam {beam.particles.dtype})")
        w, h = self.effective_resolution
        if beam is None:
            image = self.misalignment.new_zeros((int(h), int(w)))
        elif isinstance(beam, ParameterBeam):
            # bivariate normal density sampled at the pixel origins (screen.py:255-291)
            geom = self._compute_gauss_geom() if self.pixel_size.requires_grad else self._geometry("gauss_geom")
            nx, ny = self._compute_sample_counts() if self.pixel_size.requires_grad else self._geometry("sample_counts")
            image = _ops.screen_gaussian(beam.mu, beam.cov, self.misalignment, geom, nx, ny)
        elif self.method == "histogram":
            if beam.particles.dim() > 2 or beam.particle_charges.dim() > 1 or beam.energy.dim() > 0:
                raise NotImplementedError("The 'histogram' method of Screen does not support vectorization. "
                                          "Use 'cloud-in-cell' instead.")
            ex, ey = self.pixel_bin_edges
            image = _ops.hist2d(beam.particles, ex, ey, charge=beam.particle_charges,
                                survival=beam.survival_probabilities, shift=self.misalignment)
        elif self.method == "kde":
            from .. import sharding

            cx, cy = self.pixel_bin_centers
            image = _ops.kde_histogram_2d(beam.particles, cx, cy, self.kde_bandwidth, charge=beam.particle_charges,
                                          survival=beam.survival_probabilities, shift=self.misalignment,
                                          group=sharding.active_group())
        else:
            image = _ops.cic_deposit(beam.particles, (0, 2), (w, h), self.extent.reshape(2, 2),
                                     charge=beam.particle_charges, survival=beam.survival_probabilities,
                                     shift=self.misalignment, abs_charge=True, transpose_2d=True)
        from .. import sharding

        group = sharding.active_group()
        if group is not None and beam is not None and not isinstance(beam, ParameterBeam) and self.method != "kde":
            # every rank deposited its own particles: the image of ALL shards (a collective; a graph on this rank's share is
            # kept — the backward pass of the sum is the identity, sharding.sum_over_ranks). The 'kde' image sums its kernel
            # values over the ranks in front of its normalisation (_ops.kde_histogram_2d).
            image = sharding.sum_over_ranks(image.contiguous(), group)
        self.__dict__["_cached_reading"] = image
        return image

    def get_read_beam(self) -> ParticleBeam | None:
        """The beam as seen by the screen, i.e. with x, y relative to the screen centre (screen.py:196-214)."""
        if self.__dict__.get("_read_beam") is None and self._incoming_beam() is not None:
            inc = self.__dict__["_incoming"]
            zero = self.__dict__.get("_zero_misalignment")
            if zero is not None and zero[0] is self.misalignment and zero[1] == zero[0]._version \
                    and not isinstance(inc, ParameterBeam) and inc.particles.dtype == zero[0].dtype:
                # centred screen: x - 0 = x bit for bit; the recorded snapshot already is a private copy
                self.__dict__["_read_beam"] = inc
                return inc
            ref = inc.mu if isinstance(inc, ParameterBeam) else inc.particles
            tm = torch.eye(7, dtype=ref.dtype, device=ref.device).repeat(*self.misalignment.shape[:-1], 1, 1)
            tm[..., 0, 6] = -self.misalignment[..., 0]
            tm[..., 2, 6] = -self.misalignment[..., 1]
            if isinstance(inc, ParameterBeam):
                self.__dict__["_read_beam"] = inc._tracked(tm, None)  # mu_x -= mx, mu_y -= my; cov unchanged
                return self.__dict__["_read_beam"]
            shifted = _ops.apply_map(inc.particles, tm)  # x -= mx, y -= my through the apply kernel
            self.__dict__["_read_beam"] = ParticleBeam(
                shifted, inc.energy, particle_charges=inc.particle_charges,
                survival_probabilities=inc.survival_probabilities, s=inc.s, species=inc.species)
        return self.__dict__.get("_read_beam")

    def set_read_beam(self, value) -> None:
        self.__dict__["_placed"] = (self.misalignment.dtype, self.misalignment.device)
        self.__dict__["_incoming"] = value
        self.__dict__["_record"] = None
        self.__dict__["_eager"] = None
        self.__dict__["_read_beam"] = value
        self.__dict__["_cached_reading"] = None

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["resolution", "pixel_size", "binning", "misalignment", "method",
                                            "kde_bandwidth", "is_active"]
