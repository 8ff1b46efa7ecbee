"""pytorch3d.structures.Meshes, as used by the reference:
constructed from padded tensors or lists (nmr.py:152, main.py:195, 600, 699-700), then queried
through verts_packed / faces_packed / edges_packed / verts_padded / num_verts_per_mesh /
isempty / device / len()."""
import torch

from .. import _lib

# edges_packed of a batch is a sort + unique over 3*N*F pairs; the reference rebuilds Meshes every
# step from the same faces tensor, so the result is memoised on (storage, shape, version).
_EDGE_CACHE = {}


def _may_keep(t):
    """A tensor made while a hipGraph is being captured lives in the graph's private pool: it must not outlive the
    capture in a module-level table (eager code would read memory the graph reuses)."""
    return not (t.is_cuda and torch.cuda.is_current_stream_capturing())


_FACES_PACKED_CACHE = {}   # same keying and lifetime rule as _EDGE_CACHE: (padded faces tensor, V) -> packed faces
_WEIGHT_CACHE = {}         # (meshes, verts per mesh, device) -> 1 / V per packed vertex


class Meshes:
    def __init__(self, verts=None, faces=None, textures=None):
        if torch.is_tensor(verts):
            self._verts_list = list(verts.unbind(0))
            self._verts_padded = verts
        else:
            self._verts_list = list(verts)
            self._verts_padded = None
        if torch.is_tensor(faces):
            self._faces_list = list(faces.unbind(0))
            self._faces_padded = faces
        else:
            self._faces_list = list(faces)
            self._faces_padded = None
        if len(self._verts_list) != len(self._faces_list):
            raise ValueError("verts and faces must describe the same number of meshes")
        self.textures = textures
        self._cache = {}

    # ---- basics
    def __len__(self):
        return len(self._verts_list)

    @property
    def device(self):
        return self._verts_list[0].device if self._verts_list else torch.device("cpu")

    def isempty(self):
        return len(self._verts_list) == 0 or all(v.numel() == 0 for v in self._verts_list)

    def verts_list(self):
        return self._verts_list

    def faces_list(self):
        return self._faces_list

    def num_verts_per_mesh(self):
        return _lib.const([v.shape[0] for v in self._verts_list], self.device, torch.int64)

    def num_faces_per_mesh(self):
        return _lib.const([f.shape[0] for f in self._faces_list], self.device, torch.int64)

    def _equal_sized(self):
        return (len({v.shape[0] for v in self._verts_list}) == 1 and
                len({f.shape[0] for f in self._faces_list}) == 1)

    def verts_padded(self):
        if self._verts_padded is None:
            vmax = max(v.shape[0] for v in self._verts_list)
            out = self._verts_list[0].new_zeros(len(self), vmax, 3)
            for i, v in enumerate(self._verts_list):
                out[i, :v.shape[0]] = v
            self._verts_padded = out
        return self._verts_padded

    def faces_padded(self):
        if self._faces_padded is None:
            fmax = max(f.shape[0] for f in self._faces_list)
            out = self._faces_list[0].new_full((len(self), fmax, 3), -1)
            for i, f in enumerate(self._faces_list):
                out[i, :f.shape[0]] = f
            self._faces_padded = out
        return self._faces_padded

    # ---- packed views (packed vertex id = sum of the previous meshes' vertex counts + v)
    def verts_packed(self):
        if self._verts_padded is not None and self._equal_sized():
            return self._verts_padded.reshape(-1, 3)
        return torch.cat(self._verts_list, 0)

    def inv_num_verts_packed(self):
        """1 / (vertices of its mesh) per packed vertex, float32 (the weights of mesh_laplacian_smoothing); a constant
        per (batch size, mesh size) for equal-sized batches, kept."""
        if self._equal_sized():
            key = (len(self), self._verts_list[0].shape[0], str(self.device))
            w = _WEIGHT_CACHE.get(key)
            if w is None:
                w = torch.full((key[0] * key[1],), 1.0 / max(key[1], 1), dtype=torch.float32, device=self.device)
                if _may_keep(w):
                    if len(_WEIGHT_CACHE) > 16:
                        _WEIGHT_CACHE.clear()
                    _WEIGHT_CACHE[key] = w
            return w
        return 1.0 / self.num_verts_per_mesh().gather(0, self.verts_packed_to_mesh_idx()).float()

    def mesh_to_verts_packed_first_idx(self):
        n = self.num_verts_per_mesh()
        return torch.cumsum(n, 0) - n

    def verts_packed_to_mesh_idx(self):
        n = self.num_verts_per_mesh()
        total = sum(v.shape[0] for v in self._verts_list)     # known on the host: no device sync
        return torch.repeat_interleave(torch.arange(len(self), device=self.device), n, output_size=total)

    def faces_packed(self):
        if "faces_packed" not in self._cache:
            if self._faces_padded is not None and self._equal_sized():
                # batches of one topology build a new Meshes every step around the SAME faces tensor: the packed ids
                # (three small launches) are memoised on it like the edges below
                fp_ = self._faces_padded
                ck = (fp_.data_ptr(), tuple(fp_.shape), tuple(fp_.stride()), fp_._version, str(fp_.device),
                      str(fp_.dtype), self._verts_list[0].shape[0])
                hit = _FACES_PACKED_CACHE.get(ck)
                if hit is None:
                    first = self.mesh_to_verts_packed_first_idx()
                    hit = (fp_, (fp_.long() + first[:, None, None]).reshape(-1, 3))
                    if _may_keep(hit[1]):
                        if len(_FACES_PACKED_CACHE) > 16:
                            _FACES_PACKED_CACHE.clear()
                        _FACES_PACKED_CACHE[ck] = hit
                self._cache["faces_packed"] = hit[1]
                return hit[1]
            first = self.mesh_to_verts_packed_first_idx()
            if self._faces_padded is not None and self._equal_sized():
                fp = (self._faces_padded.long() + first[:, None, None]).reshape(-1, 3)
            else:
                fp = torch.cat([f.long() + first[i] for i, f in enumerate(self._faces_list)], 0)
            self._cache["faces_packed"] = fp
        return self._cache["faces_packed"]

    def edges_packed(self):
        """Unique (min, max) vertex pairs of all packed faces in lexicographic order
        (SURVEY App-A.9)."""
        if "edges_packed" not in self._cache:
            ck = None
            if self._faces_padded is not None and self._equal_sized():
                fp_ = self._faces_padded
                ck = (fp_.data_ptr(), tuple(fp_.shape), tuple(fp_.stride()), fp_._version, str(fp_.device),
                      str(fp_.dtype), self._verts_list[0].shape[0])
            hit = _EDGE_CACHE.get(ck) if ck is not None else None
            if hit is not None:
                # the entry holds its source tensor: while the key is in the table the storage stays alive, so
                # the address cannot be handed to another [N,F,3] tensor with a different topology, and any
                # tensor that matches (address, strides, shape, version) is a view of the same values
                self._cache["edges_packed"] = hit[1]
            else:
                f = self.faces_packed()
                e = torch.cat([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 0)
                e = torch.sort(e, dim=1)[0]
                V = int(self.num_verts_per_mesh().sum().item())
                h = torch.unique(e[:, 0] * V + e[:, 1], sorted=True)
                self._cache["edges_packed"] = torch.stack([h // V, h % V], 1)
                if ck is not None and _may_keep(self._cache["edges_packed"]):
                    if len(_EDGE_CACHE) > 16:
                        _EDGE_CACHE.clear()
                    _EDGE_CACHE[ck] = (fp_, self._cache["edges_packed"])
        return self._cache["edges_packed"]

    def laplacian_packed(self):
        """Uniform Laplacian, sparse [sum V, sum V]: L[i,j] = 1/deg(i) on edges, L[i,i] = -1."""
        e = self.edges_packed()
        V = self.verts_packed().shape[0]
        idx = torch.cat([e.t(), e.flip(1).t()], 1)
        ones = torch.ones(idx.shape[1], dtype=torch.float32, device=self.device)
        deg = torch.zeros(V, dtype=torch.float32, device=self.device).index_add_(0, idx[0], ones)
        val = 1.0 / deg[idx[0]]
        diag = torch.arange(V, device=self.device)
        idx = torch.cat([idx, torch.stack([diag, diag])], 1)
        val = torch.cat([val, -torch.ones(V, dtype=torch.float32, device=self.device)])
        return torch.sparse_coo_tensor(idx, val, (V, V)).coalesce()

    def update_padded(self, new_verts_padded):
        return Meshes(verts=new_verts_padded, faces=self.faces_padded(), textures=self.textures)

    def to(self, device):
        return Meshes(verts=[v.to(device) for v in self._verts_list],
                      faces=[f.to(device) for f in self._faces_list], textures=self.textures)
