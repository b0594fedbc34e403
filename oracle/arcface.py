"""CPU restatement (numpy, fp32) of the reference's face-embedding path, meerqat/image/face_recognition.py:44-102 -- TEST
INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's CPU leg may import it.

**Parity unpinned.**  The arithmetic lives in three un-vendored dependencies, none installable in the build container:
  * `arcface_torch.backbones.get_model('r50', fp16=True)` (insightface, README.rst:255-261): IResNet-50.  Restated from the published
    definition (recognition/arcface_torch/backbones/iresnet.py): stem conv3x3(3, 64) - BN - PReLU; four stages of IBasicBlock
    (BN - conv3x3 - BN - PReLU - conv3x3(stride) - BN, + identity or conv1x1(stride) - BN) with [3, 4, 14, 3] blocks of 64 / 128 /
    256 / 512 channels, each stage's first block striding by 2; BN - flatten (NCHW order) - dropout(0) - fc(25088, 512) - BN1d.
    BatchNorm eps 1e-5, eval mode.  The reference runs it under fp16 autocast; this restatement (and the HIP path, which is
    fp32-class) computes in fp32: closer to the fp32 weights than the reference itself.
  * `skimage.transform.SimilarityTransform.estimate` (Umeyama, with scale) for the 5-point alignment,
  * `cv2.warpAffine(image, M, (112, 112), borderValue=0.0)`: inverse map in double, fixed-point coordinates (AB_BITS = 10,
    INTER_BITS = 5) and bilinear weights (2^15 scale, the 2 x 2 table's sum forced to 2^15), constant border 0.
Key names follow the insightface checkpoint (`backbone.pth`): conv1.weight, bn1.*, prelu.weight, layer{1..4}.{i}.{bn1,conv1,bn2,
prelu,conv2,bn3}.*, layer{s}.0.downsample.{0,1}.*, bn2.*, fc.{weight,bias}, features.*."""
import numpy as np

LAYERS = (3, 4, 14, 3)
PLANES = (64, 128, 256, 512)
EPS = 1e-5
# insightface recognition/arcface_torch/eval_ijbc.py, as copied by the reference (face_recognition.py:33-40): 112 x 112 template
SRC = np.array([[30.2946, 51.6963], [65.5318, 51.5014], [48.0252, 71.7366], [33.5493, 92.3655], [62.7299, 92.2041]], dtype=np.float32)
SRC[:, 0] += 8.0


def seeded_state(seed=0, layers=LAYERS, num_features=512):
    """Random weights in the checkpoint's layout (fp32 numpy), scaled so that activations stay O(1) through the 50 layers."""
    rng = np.random.default_rng(seed)
    st = {}

    def conv(name, cout, cin, k):
        st[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(1.0 / (cin * k * k))).astype(np.float32)

    def bn(name, c, lo=0.6, hi=1.4):
        st[name + ".weight"] = rng.uniform(lo, hi, c).astype(np.float32)
        st[name + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        st[name + ".running_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        st[name + ".running_var"] = rng.uniform(0.6, 1.4, c).astype(np.float32)

    conv("conv1", 64, 3, 3)
    bn("bn1", 64)
    st["prelu.weight"] = rng.uniform(0.1, 0.4, 64).astype(np.float32)
    inplanes = 64
    for s, (n, planes) in enumerate(zip(layers, PLANES), start=1):
        for i in range(n):
            p = f"layer{s}.{i}"
            bn(p + ".bn1", inplanes)
            conv(p + ".conv1", planes, inplanes, 3)
            bn(p + ".bn2", planes)
            st[p + ".prelu.weight"] = rng.uniform(0.1, 0.4, planes).astype(np.float32)
            conv(p + ".conv2", planes, planes, 3)
            bn(p + ".bn3", planes, 0.15, 0.35)  # a small residual branch: the trunk stays O(1) through 24 blocks
            if i == 0:
                conv(p + ".downsample.0", planes, inplanes, 1)
                bn(p + ".downsample.1", planes)
            inplanes = planes
    bn("bn2", 512)
    st["fc.weight"] = (rng.standard_normal((num_features, 512 * 49)) * np.sqrt(1.0 / (512 * 49))).astype(np.float32)
    st["fc.bias"] = (0.1 * rng.standard_normal(num_features)).astype(np.float32)
    bn("features", num_features)
    return st


def _bn(x, st, name):  # x: [B, C, H, W] or [B, C]
    shape = (1, -1) + (1,) * (x.ndim - 2)
    a = st[name + ".weight"] / np.sqrt(st[name + ".running_var"] + np.float32(EPS))
    return (x - st[name + ".running_mean"].reshape(shape)) * a.reshape(shape).astype(np.float32) + st[name + ".bias"].reshape(shape)


def _prelu(x, w):
    return np.where(x >= 0, x, x * w.reshape(1, -1, 1, 1)).astype(np.float32)


def _conv(x, w, stride, pad):
    """x [B, C, H, W], w [O, C, k, k] -> [B, O, Ho, Wo] (fp32 im2col + sgemm)."""
    B, C, H, W = x.shape
    O, _, k, _ = w.shape
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    s = x.strides
    cols = np.lib.stride_tricks.as_strided(x, (B, C, k, k, Ho, Wo), (s[0], s[1], s[2], s[3], s[2] * stride, s[3] * stride))
    cols = np.ascontiguousarray(cols.transpose(0, 4, 5, 1, 2, 3)).reshape(B * Ho * Wo, C * k * k)
    y = cols @ w.reshape(O, -1).T
    return np.ascontiguousarray(y.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2))


def iresnet_forward(st, x, layers=LAYERS):
    """x: fp32 [B, 3, 112, 112] in [-1, 1] -> [B, 512]."""
    x = np.asarray(x, np.float32)
    x = _prelu(_bn(_conv(x, st["conv1.weight"], 1, 1), st, "bn1"), st["prelu.weight"])
    for s, n in enumerate(layers, start=1):
        for i in range(n):
            p = f"layer{s}.{i}"
            stride = 2 if i == 0 else 1
            out = _conv(_bn(x, st, p + ".bn1"), st[p + ".conv1.weight"], 1, 1)
            out = _prelu(_bn(out, st, p + ".bn2"), st[p + ".prelu.weight"])
            out = _bn(_conv(out, st[p + ".conv2.weight"], stride, 1), st, p + ".bn3")
            identity = _bn(_conv(x, st[p + ".downsample.0.weight"], stride, 0), st, p + ".downsample.1") if i == 0 else x
            x = (out + identity).astype(np.float32)
    x = _bn(x, st, "bn2").reshape(x.shape[0], -1)           # torch.flatten(x, 1) of NCHW
    x = x @ st["fc.weight"].T + st["fc.bias"]
    return _bn(x.astype(np.float32), st, "features").astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------- alignment
def umeyama(src, dst):
    """skimage.transform._geometric._umeyama(src, dst, estimate_scale=True): the 3 x 3 similarity taking src to dst (float64)."""
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
        return np.nan * T
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
    return T


def invert_affine(M):
    """cv::warpAffine's inversion of the 2 x 3 matrix (double), unless WARP_INVERSE_MAP."""
    M = np.array(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _bilinear_table():
    tab = np.zeros((32, 32, 4), np.int32)
    for fy in range(32):
        for fx in range(32):
            ax, ay = np.float32(fx) * np.float32(1 / 32), np.float32(fy) * np.float32(1 / 32)
            tx, ty = (np.float32(1) - ax, ax), (np.float32(1) - ay, ay)
            w = [int(np.clip(np.rint(np.float32(ty[k1] * tx[k2]) * np.float32(32768)), -32768, 32767)) for k1 in range(2) for k2 in range(2)]
            diff = sum(w) - 32768
            if diff:
                mk = 0
                for k in range(1, 4):
                    if (w[k] > w[mk]) if diff < 0 else (w[k] < w[mk]):
                        mk = k
                w[mk] -= diff
            tab[fy, fx] = w
    return tab


_TAB = None


def warp_affine(image, M, size=112):
    """cv2.warpAffine(np.uint8 H x W x 3, M, (size, size), borderValue=0.0) -> uint8 [size, size, 3] (INTER_LINEAR)."""
    global _TAB
    if _TAB is None:
        _TAB = _bilinear_table()
    img = np.asarray(image, np.uint8)
    H, W = img.shape[:2]
    Mi = invert_affine(M).reshape(-1)
    x = np.arange(size, dtype=np.float64)
    y = np.arange(size, dtype=np.float64)

    def sat(v):
        return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)
    adelta, bdelta = sat(Mi[0] * x * 1024), sat(Mi[3] * x * 1024)
    X0, Y0 = sat((Mi[1] * y + Mi[2]) * 1024) + 16, sat((Mi[4] * y + Mi[5]) * 1024) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    w = _TAB[Y & 31, X & 31]                                   # [size, size, 4]
    acc = np.zeros((size, size, 3), np.int64)
    for dy in range(2):
        for dx in range(2):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            px = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64) * ok[..., None]
            acc += px * w[..., dy * 2 + dx, None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def align_face(image, landmarks, size=112):
    """similarity_transform (face_recognition.py:44-52): tform.estimate(landmarks, SRC); M = params[0:2]; warpAffine."""
    T = umeyama(np.asarray(landmarks, np.float32), SRC)
    return warp_affine(image, T[0:2, :], size)


def preprocess(face):
    """get_pil_preprocessor (:65-70): ToTensor() then Normalize((0.5,) * 3, (0.5,) * 3) -> fp32 [3, H, W]."""
    t = np.asarray(face, np.uint8).astype(np.float32) / np.float32(255)
    return ((t - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
