"""CPU: the STL10 binary reader and the ToTensor + Normalize restatement on a small synthetic
file written in the dataset's published layout (uint8, per image 3 channels, each channel column-major; labels
1..10).  No dataset is downloaded or read from the reference."""
import numpy as np
import pytest
import torch

from diffusion_model_amd.data import STL10Loader, normalize_uint8, read_stl10_binary


def write_split(root, n, seed=0, name="train"):
    rng = np.random.default_rng(seed)
    imgs = rng.integers(0, 256, size=(n, 3, 96, 96), dtype=np.uint8)      # row-major (C,H,W): what a reader must return
    labels = rng.integers(0, 10, size=n).astype(np.int64)
    d = root / "stl10_binary"
    d.mkdir(parents=True, exist_ok=True)
    imgs.transpose(0, 1, 3, 2).tofile(d / f"{name}_X.bin")               # on disk every channel is column-major
    (labels + 1).astype(np.uint8).tofile(d / f"{name}_y.bin")
    return imgs, labels


def test_reader_layout_and_labels(tmp_path):
    imgs, labels = write_split(tmp_path, 7)
    x, y = read_stl10_binary(str(tmp_path), "train")
    assert x.dtype == np.uint8 and x.shape == (7, 3, 96, 96) and np.array_equal(x, imgs)
    assert y.dtype == np.int64 and np.array_equal(y, labels)
    x2, _ = read_stl10_binary(str(tmp_path / "stl10_binary"), "train")   # the binary directory itself is accepted too
    assert np.array_equal(x2, imgs)


def test_reader_rejects_bad_files(tmp_path):
    with pytest.raises(FileNotFoundError):
        read_stl10_binary(str(tmp_path), "train")
    write_split(tmp_path, 3)
    with open(tmp_path / "stl10_binary" / "train_X.bin", "ab") as f:
        f.write(b"\x00" * 5)
    with pytest.raises(ValueError):
        read_stl10_binary(str(tmp_path), "train")


def test_normalize_matches_totensor_normalize():
    u = torch.arange(256, dtype=torch.uint8)
    ref = (u.float().div(255) - 0.5) / 0.5      # ToTensor(): /255 in fp32; Normalize(0.5, 0.5): (x - mean) / std
    out = normalize_uint8(u)
    assert torch.equal(out, ref) and float(out.min()) == -1.0 and float(out.max()) == 1.0


def test_loader_epochs_shuffle_and_rank_shards(tmp_path):
    imgs, labels = write_split(tmp_path, 37, seed=3)
    ld = STL10Loader(str(tmp_path), "train", batch_size=8, shuffle=True, seed=5)
    assert len(ld) == 5
    ref = normalize_uint8(torch.from_numpy(imgs))
    key = lambda r: tuple(r[:, 0, 0].tolist()) + tuple(r[:, 5, 7].tolist())  # noqa: E731  (a fingerprint per image)
    index = {key(r): i for i, r in enumerate(ref)}
    seen = []
    for x, y in ld:
        assert x.dtype == torch.float32 and x.shape[1:] == (3, 96, 96) and y.dtype == torch.int64
        for xi, yi in zip(x, y):  # every image comes back normalised and with its own label
            i = index[key(xi)]
            assert torch.equal(xi, ref[i]) and int(yi) == int(labels[i])
            seen.append(i)
    assert sorted(seen) == list(range(37))                    # one epoch = every image once
    assert seen != sorted(seen)                               # ... in shuffled order
    again = [index[key(xi)] for x, _ in STL10Loader(str(tmp_path), "train", batch_size=8, shuffle=True, seed=5) for xi in x]
    assert again == seen                                      # the order is a function of the seed
    # two ranks partition one epoch order without overlap
    a = STL10Loader(str(tmp_path), "train", batch_size=64, shuffle=True, seed=9, rank=0, world_size=2)
    b = STL10Loader(str(tmp_path), "train", batch_size=64, shuffle=True, seed=9, rank=1, world_size=2)
    xa, _ = next(iter(a)); xb, _ = next(iter(b))
    # 37 images over 2 ranks: every rank runs the same number of samples/batches (19; a rank with one batch more would
    # hang in its all-reduce), so the epoch order is padded by wrapping around -- exactly one image is seen twice
    assert xa.shape[0] == xb.shape[0] == 19 and len(a) == len(b) == 1
    got = [index[key(r)] for r in torch.cat([xa, xb])]
    assert sorted(set(got)) == list(range(37)) and len(got) == 38
    ga, gb = [index[key(r)] for r in xa], [index[key(r)] for r in xb]
    assert len(set(ga)) == 19 and len(set(gb)) == 19 and len(set(ga) & set(gb)) == 1
    # an evenly divisible split is a partition without overlap
    write_split(tmp_path / "even", 40, seed=4)
    la = [STL10Loader(str(tmp_path / "even"), "train", batch_size=4, shuffle=True, seed=2, rank=r, world_size=4) for r in range(4)]
    assert len({len(l) for l in la}) == 1 and len(la[0]) == 3
    ref_e = normalize_uint8(torch.from_numpy(read_stl10_binary(str(tmp_path / "even"), "train")[0]))
    index_e = {key(r): i for i, r in enumerate(ref_e)}
    per_rank = [[index_e[key(xi)] for x, _ in l for xi in x] for l in la]
    assert sorted(sum(per_rank, [])) == list(range(40))


def test_rank_loaders_stay_in_lockstep_across_monitoring_draws(tmp_path):
    """main_stl10.train() pulls one monitoring batch per epoch with next(iter(loader)).  iter() advances the loader's
    epoch-order generator, so the draw must happen on EVERY rank (train() does; only rank 0 uses the batch): then the next
    epoch's order[rank::world] shards still partition the epoch.  A draw on one rank alone breaks the partition."""
    write_split(tmp_path, 40, seed=6)
    ref = normalize_uint8(torch.from_numpy(read_stl10_binary(str(tmp_path), "train")[0]))
    key = lambda r: tuple(r[:, 0, 0].tolist()) + tuple(r[:, 5, 7].tolist())  # noqa: E731
    index = {key(r): i for i, r in enumerate(ref)}
    mk = lambda r: STL10Loader(str(tmp_path), "train", batch_size=5, shuffle=True, seed=3, rank=r, world_size=2)  # noqa: E731
    epoch = lambda ld: [index[key(xi)] for x, _ in ld for xi in x]  # noqa: E731
    a, b = mk(0), mk(1)
    for _ in range(2):     # two epochs, each followed by the monitoring draw on both ranks
        ea, eb = epoch(a), epoch(b)
        assert sorted(ea + eb) == list(range(40))
        next(iter(a)); next(iter(b))
    assert sorted(epoch(a) + epoch(b)) == list(range(40))
    a, b = mk(0), mk(1)
    epoch(a), epoch(b)
    next(iter(a))          # rank 0 alone (what train() must not do): the generators part ways
    assert sorted(epoch(a) + epoch(b)) != list(range(40))


def test_loader_exact_batches_unshuffled(tmp_path):
    imgs, labels = write_split(tmp_path, 10, seed=1)
    ld = STL10Loader(str(tmp_path), "train", batch_size=4, shuffle=False, drop_last=True)
    assert len(ld) == 2
    batches = list(ld)
    assert torch.equal(batches[0][0], normalize_uint8(torch.from_numpy(imgs[:4])))
    assert torch.equal(batches[1][1], torch.from_numpy(labels[4:8]))


def test_latent_loader_matches_reference_dataset(tmp_path):
    """LatentDataset (reference utils.py:70-91): sample i = latents[i] * scaling_factor; loader = shuffled batches of those,
    sharded by rank like STL10Loader."""
    import torch
    from diffusion_model_amd.data import LatentLoader
    g = torch.Generator().manual_seed(0)
    lat, lab = torch.randn(10, 4, 8, 8, generator=g), torch.randint(0, 10, (10,), generator=g)
    path = tmp_path / "stl10_latents.pt"
    torch.save({"latents": lat, "labels": lab}, path)
    ld = LatentLoader(str(path), batch_size=4, shuffle=False)
    assert len(ld) == 3
    xs, ys = zip(*list(ld))
    assert [x.shape[0] for x in xs] == [4, 4, 2]
    assert torch.equal(torch.cat(xs), lat * 0.18215) and torch.equal(torch.cat(ys), lab)
    x3, y3 = ld[3]
    assert torch.equal(x3, lat[3] * 0.18215) and int(y3) == int(lab[3])
    # two ranks see disjoint halves of one shuffled epoch
    a = LatentLoader(str(path), batch_size=16, seed=5, rank=0, world_size=2, scaling_factor=1.0)
    b = LatentLoader(str(path), batch_size=16, seed=5, rank=1, world_size=2, scaling_factor=1.0)
    xa, xb = next(iter(a))[0], next(iter(b))[0]
    both = torch.cat([xa, xb])
    assert both.shape[0] == 10
    assert sorted(float(v) for v in both.flatten(1).sum(1)) == pytest.approx(sorted(float(v) for v in lat.flatten(1).sum(1)))
