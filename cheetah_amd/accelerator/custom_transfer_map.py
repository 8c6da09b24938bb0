"""CustomTransferMap (mirror of cheetah/accelerator/custom_transfer_map.py:33-114)."""

from __future__ import annotations

import torch

from .element import Element


class CustomTransferMap(Element):
    """Carrier of a user-supplied or merged (…,7,7) first-order map."""

    supported_tracking_methods = ["linear"]

    def __init__(self, predefined_transfer_map, length=None, name=None, sanitize_name=None, metadata=None,
                 device=None, dtype=None) -> None:
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, device=device, dtype=dtype)
        if length is not None:
            self.length = length
        assert predefined_transfer_map.shape[-2:] == (7, 7)
        assert (predefined_transfer_map[..., -1, :-2] == 0.0).all() and (
            predefined_transfer_map[..., -1, -1] == 1.0
        ).all(), "The seventh row of the transfer map must be [0, 0, 0, 0, 0, 0, 1]."
        self.register_buffer_or_parameter("predefined_transfer_map", predefined_transfer_map)

    @classmethod
    def from_merging_elements(cls, elements, incoming_beam) -> "CustomTransferMap":
        """Merge consecutive skippable elements into one map (custom_transfer_map.py:60-109) with the
        `chx_compose_maps` kernel."""
        from .. import _ops

        assert all(e.is_skippable for e in elements), \
            "Combining the elements in a Segment that is not skippable will result in incorrect tracking results."
        energy, species = incoming_beam.energy, incoming_beam.species
        maps = [e.first_order_transfer_map(energy, species) for e in elements]
        batch_shape = torch.broadcast_shapes(energy.shape, *[m.shape[:-2] for m in maps])
        tm = _ops.compose_maps(maps, batch_shape, maps[0].dtype, maps[0].device)
        length = sum(e.length for e in elements)
        return cls(tm, length=length, name="combined_" + "_".join(e.name for e in elements), sanitize_name=False,
                   device=tm.device, dtype=tm.dtype)

    def first_order_transfer_map(self, energy, species) -> torch.Tensor:
        return self.predefined_transfer_map

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "predefined_transfer_map"]
