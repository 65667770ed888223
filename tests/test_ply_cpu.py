"""-m "not gpu": the on-disk point-cloud format (S3Gaussian/scene/gaussian_model.py:245-298,378-425) -- byte layout of the file
written by emd_amd.gaussian_model (the PLY `plyfile` produces for the reference's attribute list) and the reader."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_ply_bytes_and_reader(tmp_path):
    from emd_amd.gaussian_model import GaussianModel, read_ply, write_ply
    z = np.load(os.path.join(G, "s3g_surgery.npz"))
    m = GaussianModel(device="cpu")
    P = lambda k: torch.nn.Parameter(torch.tensor(z[f"in_{k}"]))
    for n, attr in m._ATTR.items():
        setattr(m, attr, P(n))
    names = m.construct_list_of_attributes()
    assert names == [str(a) for a in z["attributes"]] and len(names) == 66          # the reference's list, verbatim
    path = str(tmp_path / "a" / "pc.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    N = z["in_xyz"].shape[0]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N + "".join(f"property float {n}\n" for n in names) + "end_header\n").encode()
    assert raw.startswith(header) and len(raw) == len(header) + N * 66 * 4
    body = np.frombuffer(raw[len(header):], dtype="<f4").reshape(N, 66)
    np.testing.assert_array_equal(body[:, 0:3], z["in_xyz"])
    assert np.all(body[:, 3:6] == 0)                                                # normals
    np.testing.assert_array_equal(body[:, 6:9], z["in_f_dc"].transpose(0, 2, 1).reshape(N, 3))
    np.testing.assert_array_equal(body[:, 9:54], z["in_f_rest"].transpose(0, 2, 1).reshape(N, 45))      # channel-major, as the reference flattens
    np.testing.assert_array_equal(body[:, 54], z["in_opacity"][:, 0])
    np.testing.assert_array_equal(body[:, 55:58], z["in_scaling"])
    np.testing.assert_array_equal(body[:, 58:62], z["in_rotation"])
    np.testing.assert_array_equal(body[:, 62:66], z["in_embedding"])
    d = read_ply(path)
    assert list(d) == names
    m2 = GaussianModel(device="cpu")
    m2.load_ply(path)
    for n, attr in m._ATTR.items():
        assert torch.equal(getattr(m2, attr).detach(), getattr(m, attr).detach()), n
    # a 62-column file (what the reference's save_ply would write without embeddings) and an ascii file load as well
    write_ply(str(tmp_path / "b.ply"), names[:62], body[:, :62])
    m3 = GaussianModel(device="cpu")
    m3.load_ply(str(tmp_path / "b.ply"))
    assert m3._embedding.shape == (N, 0) and torch.equal(m3._rotation.detach(), m._rotation.detach())
    with open(tmp_path / "c.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        for r in body[:2]:
            f.write(" ".join(repr(float(v)) for v in r) + "\n")
    d3 = read_ply(str(tmp_path / "c.ply"))
    np.testing.assert_allclose(np.stack([d3[n] for n in names], 1), body[:2], rtol=1e-7)
