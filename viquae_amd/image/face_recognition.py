"""Host-side mirror of ``meerqat.image.face_recognition`` (meerqat/image/face_recognition.py): ArcFace embeddings of the faces an
upstream detector (``meerqat.image.face_detection``, out of scope) left as 5-point ``face_landmarks`` in the dataset.

Same names and call shapes as the reference for the hot-path surface:

* ``SRC``, ``ARCFACE_PATH``, ``PRETRAINED_MODELS``            :29-40
* ``similarity_transform(image, landmarks, src, tform)``        :44-52   (one face, on the host: PIL in, PIL out)
* ``from_pretrained(model_name='r50', fp16=True, train=False)`` :55-61   -> :class:`viquae_amd.arcface.ArcFaceR50`
* ``get_pil_preprocessor()``                                    :64-70
* ``compute_face_embedding(batch, model, preprocessor, tform, max_n_faces=1, image_key='image')``   :72-102
* ``dataset_compute_face_embedding(dataset_path, map_kwargs, pretrained_kwargs, fn_kwargs)``        :105-112

What differs underneath: the reference aligns every face on the host (scikit-image's Umeyama estimate, cv2.warpAffine, PIL,
torchvision's ToTensor / Normalize), one at a time; here only the 2 x 3 matrices are estimated on the host (numpy, a few hundred
flops per face) and ``compute_face_embedding`` cuts, scales and normalises ALL faces of a batch in one kernel
(``mq_warp_affine_faces_f32``: OpenCV's fixed-point bilinear arithmetic + ToTensor + Normalize) from the decoded images packed in
one device buffer, then runs :class:`ArcFaceR50`.  ``fp16`` is accepted and ignored: the HIP model is fp32-class throughout.

Neither scikit-image, OpenCV nor arcface_torch is vendored by the reference or installable here: parity unpinned
(oracle/arcface.py restates the published algorithms; tests/test_arcface_gpu.py)."""
import warnings

import numpy as np

from ..data import loading
from ..data.loading import DATA_ROOT_PATH

ARCFACE_PATH = DATA_ROOT_PATH / "arcface"
PRETRAINED_MODELS = {
    "r50": ARCFACE_PATH / "ms1mv3_arcface_r50_fp16" / "backbone.pth"
}
# insightface recognition/arcface_torch/eval_ijbc.py, as the reference copies it (:33-40)
SRC = np.array([
    [30.2946, 51.6963],
    [65.5318, 51.5014],
    [48.0252, 71.7366],
    [33.5493, 92.3655],
    [62.7299, 92.2041]], dtype=np.float32)
SRC[:, 0] += 8.0
IMAGE_SIZE = 112


class SimilarityTransform:
    """``skimage.transform.SimilarityTransform`` as far as the reference uses it: ``estimate(src, dst)`` (Umeyama's closed form,
    with scale, float64) and ``params`` (3 x 3)."""

    def __init__(self):
        self.params = np.eye(3, dtype=np.float64)

    def estimate(self, src, dst):
        src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
        num, dim = src.shape
        src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
        src_demean, dst_demean = src - src_mean, dst - dst_mean
        A = dst_demean.T @ src_demean / num
        d = np.ones((dim,), dtype=np.float64)
        if np.linalg.det(A) < 0:
            d[dim - 1] = -1
        T = np.eye(dim + 1, dtype=np.float64)
        U, S, V = np.linalg.svd(A)
        rank = np.linalg.matrix_rank(A)
        if rank == 0:
            self.params = np.nan * T
            return False
        if rank == dim - 1:
            if np.linalg.det(U) * np.linalg.det(V) > 0:
                T[:dim, :dim] = U @ V
            else:
                s = d[dim - 1]
                d[dim - 1] = -1
                T[:dim, :dim] = U @ np.diag(d) @ V
                d[dim - 1] = s
        else:
            T[:dim, :dim] = U @ np.diag(d) @ V
        scale = 1.0 / src_demean.var(axis=0).sum() * (S @ d)
        T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
        T[:dim, :dim] *= scale
        self.params = T
        return True


def _invert_affine(M):
    """cv::warpAffine inverts the 2 x 3 matrix it is given (no WARP_INVERSE_MAP) in double precision."""
    M = np.array(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def face_matrices(arg):
    """(landmarks of one image, max_n_faces) -> (number of faces kept, [the inverted 2 x 3 matrix of each face as cv::warpAffine
    uses it]): ``tform.estimate(landmark, SRC)`` + the inversion, per face like ``compute_face_embedding`` does it -- the same numpy
    calls, so the same bits.  The pipelined job runs it in the decode workers (~50 us of small numpy calls per face)."""
    landmarks, max_n_faces = arg
    lm = np.array(landmarks[:max_n_faces], dtype=np.float32)
    tform = SimilarityTransform()
    mats = []
    for landmark in lm:
        tform.estimate(landmark, SRC)
        mats.append(_invert_affine(tform.params[0:2, :]))
    return int(lm.shape[0]), mats


def align_faces_device(images, faces, device, image_size=IMAGE_SIZE):
    """``images``: list of uint8 [H, W, 3] arrays; ``faces``: list of (image index, 2 x 3 matrix M as tform.params[0:2]).
    -> fp32 CUDA tensor [len(faces), 3, size, size]: warpAffine + ToTensor + Normalize(0.5, 0.5) of every face, one kernel."""
    import torch
    from .. import _lib
    lib = _lib.load()
    arrays = [np.ascontiguousarray(im, np.uint8) for im in images]
    offsets = np.zeros(len(arrays), np.int64)
    hw = np.zeros((len(arrays), 2), np.int32)
    total = 0
    for i, a in enumerate(arrays):
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("expected RGB images [H, W, 3]")
        offsets[i], hw[i] = total, a.shape[:2]
        total += a.size
    packed = np.empty(total, np.uint8)
    for a, o in zip(arrays, offsets):
        packed[o:o + a.size] = a.reshape(-1)
    minv = np.stack([_invert_affine(M) for _, M in faces]).reshape(-1)
    owner = np.array([i for i, _ in faces], np.int32)
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    packed_d, off_d, hw_d, own_d, minv_d = t(packed), t(offsets), t(hw.reshape(-1)), t(owner), t(minv)
    out = torch.empty((len(faces), 3, image_size, image_size), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mq_warp_affine_faces_f32(packed_d.data_ptr(), off_d.data_ptr(), hw_d.data_ptr(), own_d.data_ptr(), minv_d.data_ptr(),
                                                len(faces), image_size, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "mq_warp_affine_faces_f32")
    return out


def similarity_transform(image, landmarks, src, tform, image_size=IMAGE_SIZE):
    """One face, as the reference does it (:44-52): estimate, warp to ``image_size``, return a PIL image.  The warp runs on the
    device (there is no OpenCV here); ``compute_face_embedding`` does all faces of a batch at once instead."""
    from PIL import Image
    tform.estimate(landmarks, src)
    M = tform.params[0:2, :]
    t = align_faces_device([np.array(image, dtype=np.uint8)], [(0, M)], "cuda", image_size)[0]
    face = ((t * 0.5 + 0.5) * 255.0).round().clamp(0, 255).to("cpu").numpy().astype(np.uint8).transpose(1, 2, 0)
    return Image.fromarray(face)


def from_pretrained(model_name="r50", fp16=True, train=False):
    """:55-61.  ``fp16`` (the reference's autocast switch) is accepted and ignored; ``train=True`` is not supported (inference)."""
    from ..arcface import ArcFaceR50
    from ..utils import device
    if model_name not in PRETRAINED_MODELS:
        raise KeyError(model_name)
    if train:
        raise NotImplementedError("the HIP face encoder is inference-only")
    return ArcFaceR50.from_pretrained(PRETRAINED_MODELS[model_name]).to(device=device)


def get_pil_preprocessor():
    """ToTensor() + Normalize((0.5,) * 3, (0.5,) * 3) of a PIL image -> fp32 tensor [3, H, W] (:64-70), without torchvision."""
    import torch

    def preprocess(face):
        t = torch.from_numpy(np.asarray(face, np.uint8).astype(np.float32) / np.float32(255)).permute(2, 0, 1)
        return (t - 0.5) / 0.5
    return preprocess


def compute_face_embedding(batch, model, preprocessor, tform, max_n_faces=1, image_key="image"):
    """:72-102: images without detected faces (``face_landmarks`` None) or unreadable get None; the others a [n_faces, 512] array
    of the embeddings of their first ``max_n_faces`` faces.  ``preprocessor`` is part of the reference's signature; the device path
    fuses its arithmetic into the alignment kernel (a model that is not on a GPU raises: no CPU fallback)."""
    output = []
    images, faces, not_None_values_indices = [], [], []
    for i, (image, landmarks) in enumerate(zip(batch[image_key], batch["face_landmarks"])):
        output.append(None)  # overwritten for images with faces
        if landmarks is not None:
            image = loading.load_image(image)
            if image is None:  # the reference crashes here on an unreadable image (np.array(None)); a warning has been issued
                continue
            landmarks = np.array(landmarks[:max_n_faces], dtype=np.float32)
            images.append(np.asarray(image, np.uint8))
            for landmark in landmarks:
                tform.estimate(landmark, SRC)
                faces.append((len(images) - 1, tform.params[0:2, :].copy()))
            not_None_values_indices.append((i, landmarks.shape[0]))
    if not faces:
        batch["face_embedding"] = output
        return batch
    import torch
    first = next(iter(model.parameters()), None) if hasattr(model, "parameters") else None
    first = first if first is not None else next(iter(model.buffers()), None)
    if first is None or not first.is_cuda:
        from .. import _lib
        raise _lib.MeerqatHipError("viquae_amd face embedding runs on MI355X only: move the model to a GPU (no CPU fallback)")
    with torch.no_grad():
        pixel_values = align_faces_device(images, faces, first.device)
        not_None_output = model(pixel_values).cpu().numpy()
    j = 0
    for i, n_faces in not_None_values_indices:
        output[i] = not_None_output[j: j + n_faces]
        j += n_faces
    batch["face_embedding"] = output
    return batch


def dataset_compute_face_embedding(dataset_path, map_kwargs={}, pretrained_kwargs={}, fn_kwargs={}):
    """:105-112 (the reference overwrites the dataset in place; recent ``datasets`` refuses that: written next to it, then swapped).
    ``compute_face_embedding``'s own arguments found in ``map_kwargs`` -- the shipped experiments/face_recognition/config.json puts
    ``max_n_faces`` there, which ``Dataset.map`` rejects -- are moved to ``fn_kwargs``.  A plain job runs software-pipelined
    (viquae_amd.pipeline.FaceEmbedPipeline: decode workers, JPEG scans finished on the GPU, alignment of batch i + 1 behind the
    ArcFace forward of batch i); ``MQ_EMBED_PIPELINE=0`` maps the serial ``compute_face_embedding``."""
    from datasets import load_from_disk
    from .embedding import _save
    from .decode_pool import early_pool
    from ..pipeline import _map_batch_size, face_pipeline_or_none
    map_kwargs, fn_kwargs = dict(map_kwargs), dict(fn_kwargs)
    for k in ("max_n_faces", "image_key"):
        if k in map_kwargs:
            fn_kwargs.setdefault(k, map_kwargs.pop(k))
    dataset = load_from_disk(dataset_path)
    # the decode workers are forked FIRST, while this process owns no page-locked memory (decode_pool.py)
    workers = early_pool(None, _map_batch_size(map_kwargs)) if _map_batch_size(map_kwargs) is not None else None
    model = from_pretrained(**pretrained_kwargs)
    pipe = face_pipeline_or_none(dataset, map_kwargs, model=model, decode_pool=workers, **fn_kwargs)
    if pipe is None and workers is not None:
        workers.close()
    if "new_fingerprint" not in map_kwargs:  # never pickle the model for a hash
        from datasets.fingerprint import generate_random_fingerprint
        map_kwargs["new_fingerprint"] = generate_random_fingerprint()
    if pipe is not None:
        try:
            dataset = dataset.map(pipe.embed, batched=True, with_indices=True, **map_kwargs)
        finally:
            pipe.close()
            dataset_compute_face_embedding.last_pipeline_stats = dict(pipe.stats)
    else:
        dataset_compute_face_embedding.last_pipeline_stats = None
        fn_kwargs = dict(fn_kwargs, model=model, preprocessor=get_pil_preprocessor(), tform=SimilarityTransform())
        dataset = dataset.map(compute_face_embedding, batched=True, fn_kwargs=fn_kwargs, **map_kwargs)
    return _save(dataset, dataset_path, dataset_path)


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser(description="ArcFace embeddings of the faces of a dataset (python -m meerqat.image.face_recognition <dataset> [<config>] [--disable_caching])")
    ap.add_argument("dataset")
    ap.add_argument("config", nargs="?")
    ap.add_argument("--disable_caching", action="store_true")
    a = ap.parse_args()
    if a.disable_caching:
        import datasets
        datasets.disable_caching()
    dataset_compute_face_embedding(a.dataset, **(json.load(open(a.config)) if a.config else {}))
