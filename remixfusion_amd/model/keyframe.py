"""KeyFrameDatabase: per-keyframe ray store + uniform global ray sampling
(reference model/keyframe.py:5-96).  Rays are (dir3, rgb3, depth1).  Unlike the reference the
store lives on the mapping device (HBM) so sampling is a device gather, not a CPU gather +
H2D copy per iteration; Python's ``random`` still draws the indices so a
seeded run picks the same rays as the reference would."""
from __future__ import annotations

import random

import torch

from .. import _lib


class KeyFrameDatabase(object):
    def __init__(self, config, H, W, num_kf, num_rays_to_save, device, num_frame=None) -> None:
        self.config = config
        self.keyframes = {}
        self.device = device
        # mapping.device_sampling: draw ray indices on the device (rfx_random_subset)
        # instead of Python's random.sample on the host (reference behaviour, O(k) python per draw)
        self.device_sampling = bool(config["mapping"].get("device_sampling", False))
        self.rays = torch.zeros((num_kf, num_rays_to_save, 7), device=device)
        self.num_rays_to_save = num_rays_to_save
        self.frame_ids = None
        # device twin of frame_ids, pre-sized: appended to with scalar fills, so adding a keyframe neither copies
        # host memory to the device nor drains the stream
        self.frame_ids_dev = torch.zeros((num_kf,), dtype=torch.int64, device=device)
        self.H, self.W = H, W
        self.kf_poses = torch.zeros((num_kf, 4, 4))
        self.kf_fuse_poses = torch.zeros((num_kf, 4, 4))
        self.kf_error = torch.zeros((num_kf), device=device)
        self.kf_error_cnt = torch.zeros((num_kf), device=device)
        if num_frame is not None:
            self.all_fuse_pose = torch.zeros((num_frame, 4, 4), device=device)

    def __len__(self):
        return len(self.frame_ids)

    def get_length(self):
        return self.__len__()

    def _choose(self, population: int, k: int, device) -> torch.Tensor:
        """k distinct indices out of range(population)."""
        if self.device_sampling:
            return _lib.random_subset(population, k, device)
        return torch.as_tensor(random.sample(range(0, population), k), device=device)

    def sample_single_keyframe_rays(self, rays, option="random", first=False):
        """rays [1, H*W, 7] -> [1, num_rays_to_save, 7] (or [num_rays_to_save, 7] for filter_depth)."""
        rays_valid = None
        if option == "random":
            idx_t = self._choose(self.H * self.W, self.num_rays_to_save, rays.device)
        elif option == "filter_depth" and self.device_sampling and rays.is_cuda:
            return self._filter_depth_on_device(rays, first)
        elif option == "filter_depth":
            valid = (rays[..., -1] > 0.0) & (rays[..., -1] <= self.config["cam"]["depth_trunc"])
            rays_valid = rays[valid, :]
            if len(rays_valid) > self.num_rays_to_save:
                idx_t = self._choose(len(rays_valid), self.num_rays_to_save, rays.device)
            else:
                # too few valid-depth rays: fall back to uniform sampling over the frame.  (The
                # reference intends this too but its `option == "random"` at :42 is a comparison,
                # so it would index rays_valid out of range; SURVEY.md appendix D.)
                idx_t = self._choose(self.H * self.W, self.num_rays_to_save, rays.device)
                option = "random"
        else:
            raise NotImplementedError()
        if option == "random" or first:
            return rays[:, idx_t]
        return rays_valid[idx_t, :]

    def _filter_depth_on_device(self, rays, first):
        """option 'filter_depth' (reference :37-52) without bringing the number of valid rays to the host: the boolean-mask
        gather `rays[valid]` and `len(rays_valid)` of the reference wait for the device, i.e. for everything the mapper has
        queued (5.6 ms per keyframe at scene0000 sizes, the frame loop's only synchronisation).  The k indices are drawn
        among the valid rays by rfx_random_subset_dev (or among all rays when k or fewer are valid), and the j-th valid ray
        is found by a search in the running count of valid rays.  Same distribution; the reference's quirk for the first
        frame (indices drawn among the valid rays but applied to ALL rays) is kept."""
        import random as _random
        r = rays.reshape(-1, rays.shape[-1])
        valid = (r[:, -1] > 0.0) & (r[:, -1] <= self.config["cam"]["depth_trunc"])
        running = valid.to(torch.int64).cumsum(0)
        k = self.num_rays_to_save
        j = torch.empty(k, dtype=torch.int64, device=r.device)
        fb = torch.empty(1, dtype=torch.int32, device=r.device)
        _lib.check(_lib.load().rfx_random_subset_dev(_random.getrandbits(64), running[-1:].data_ptr(), r.shape[0], k, j.data_ptr(),
                                                     fb.data_ptr(), _lib.stream_ptr(r.device)), "rfx_random_subset_dev")
        if first:
            return rays[:, j] if rays.dim() == 3 else r[j]
        idx = torch.where(fb.bool(), j, torch.searchsorted(running, j + 1))
        return r[idx]

    def attach_ids(self, frame_ids):
        n0 = 0 if self.frame_ids is None else len(self.frame_ids)
        self.frame_ids = frame_ids if self.frame_ids is None else torch.cat([self.frame_ids, frame_ids], dim=0)
        for j, v in enumerate(frame_ids.tolist()):
            self.frame_ids_dev[n0 + j].fill_(int(v))

    def add_keyframe(self, batch, filter_depth=False):
        first = bool(batch["frame_id"] == 0)
        rays = batch.get("_rays7")                      # built once per mapper step (Mapper.step)
        if rays is None:
            rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1)
        rays = rays.reshape(1, -1, rays.shape[-1]).to(self.device)
        rays = self.sample_single_keyframe_rays(rays, "filter_depth" if filter_depth else "random", first=first)
        fid = batch["frame_id"]
        if not isinstance(fid, torch.Tensor):
            fid = torch.tensor([fid])
        self.attach_ids(fid.reshape(-1).cpu())
        self.rays[len(self.frame_ids) - 1] = rays

    def sample_global_rays(self, bs):
        num_kf = self.__len__()
        if self.device_sampling:
            idxs = _lib.random_subset(num_kf * self.num_rays_to_save, bs, self.rays.device)
            sample_rays = self.rays[:num_kf].reshape(-1, 7)[idxs]
            frame_ids = self.frame_ids_dev[idxs // self.num_rays_to_save]
            return sample_rays, frame_ids
        idxs = torch.tensor(random.sample(range(num_kf * self.num_rays_to_save), bs))
        sample_rays = self.rays[:num_kf].reshape(-1, 7)[idxs.to(self.rays.device)]
        frame_ids = self.frame_ids[idxs // self.num_rays_to_save]
        return sample_rays, frame_ids
