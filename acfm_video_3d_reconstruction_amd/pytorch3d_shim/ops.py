"""pytorch3d.ops.SubdivideMeshes (multiframe/main.py:196-199); semantics: SURVEY App-A.8."""
import torch

from .structures import Meshes


class SubdivideMeshes(torch.nn.Module):
    """One 1->4 subdivision: a new vertex at every edge midpoint (appended after the originals
    in edges_packed order); face (v0,v1,v2) with edge ids e0=v1v2, e1=v0v2, e2=v0v1 becomes
    (v0,e2,e1), (v1,e0,e2), (v2,e1,e0), (e0,e1,e2), concatenated as [all f0; f1; f2; f3]."""

    def __init__(self, meshes=None):
        super().__init__()
        self.precomputed = False
        if meshes is not None:
            if len(meshes) != 1:
                raise ValueError("Mesh can only have one mesh.")
            self.register_buffer("_subdivided_faces", self._subdivide_faces(meshes))
            self.precomputed = True

    @staticmethod
    def _subdivide_faces(mesh):
        faces = mesh.faces_packed()
        edges = mesh.edges_packed()
        V = mesh.verts_packed().shape[0]
        key = edges[:, 0] * V + edges[:, 1]

        def eid(a, b):
            k = torch.minimum(a, b) * V + torch.maximum(a, b)
            return torch.searchsorted(key, k) + V

        v0, v1, v2 = faces[:, 0], faces[:, 1], faces[:, 2]
        e0, e1, e2 = eid(v1, v2), eid(v0, v2), eid(v0, v1)
        f0 = torch.stack([v0, e2, e1], 1)
        f1 = torch.stack([v1, e0, e2], 1)
        f2 = torch.stack([v2, e1, e0], 1)
        f3 = torch.stack([e0, e1, e2], 1)
        return torch.cat([f0, f1, f2, f3], 0)

    def forward(self, meshes, feats=None):
        if feats is not None:
            raise NotImplementedError("per-vertex features are not used by the reference")
        out_v, out_f = [], []
        for i in range(len(meshes)):
            m = Meshes(verts=[meshes.verts_list()[i]], faces=[meshes.faces_list()[i]])
            faces = self._subdivided_faces if self.precomputed else self._subdivide_faces(m)
            e = m.edges_packed()
            v = m.verts_packed()
            out_v.append(torch.cat([v, 0.5 * (v[e[:, 0]] + v[e[:, 1]])], 0))
            out_f.append(faces)
        return Meshes(verts=out_v, faces=out_f)
