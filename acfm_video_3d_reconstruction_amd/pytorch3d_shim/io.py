"""pytorch3d.io.load_obj as used at multiframe/main.py:159-160 and predictor.py:64:
returns (verts [V,3] f32, faces namedtuple with .verts_idx [F,3] i64, aux)."""
from collections import namedtuple

import torch

Faces = namedtuple("Faces", "verts_idx normals_idx textures_idx materials_idx")
Properties = namedtuple("Properties", "normals verts_uvs material_colors texture_images texture_atlas")


def load_obj(f, load_textures=False, **kwargs):
    """Accepts 'f a b c' and 'f a/ta/na ...' records (bird.obj uses the latter);
    polygons are fan-triangulated like PyTorch3D."""
    verts, vts, faces, tfaces = [], [], [], []
    fh = open(f) if isinstance(f, str) else f
    try:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vts.append([float(x) for x in p[1:3]])
            elif p[0] == "f":
                idx = [q.split("/") for q in p[1:]]
                vi = [int(q[0]) for q in idx]
                vi = [i - 1 if i > 0 else len(verts) + i for i in vi]
                ti = [int(q[1]) - 1 if len(q) > 1 and q[1] else -1 for q in idx]
                for k in range(1, len(vi) - 1):
                    faces.append([vi[0], vi[k], vi[k + 1]])
                    tfaces.append([ti[0], ti[k], ti[k + 1]])
    finally:
        if isinstance(f, str):
            fh.close()
    v = torch.tensor(verts, dtype=torch.float32).reshape(-1, 3)
    fi = torch.tensor(faces, dtype=torch.int64).reshape(-1, 3)
    ti = torch.tensor(tfaces, dtype=torch.int64).reshape(-1, 3)
    uv = torch.tensor(vts, dtype=torch.float32).reshape(-1, 2) if vts else None
    return v, Faces(fi, torch.full_like(fi, -1), ti, torch.full((fi.shape[0],), -1, dtype=torch.int64)), \
        Properties(None, uv, None, None, None)
