"""Oracle for the similarity filter.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch


def ref_mean_and_thres(ref_feat):
    """get_ref_mean_vec_and_thres_from_imgs arithmetic
    (similar_face_filtering/filter_faces_using_reference.py:85-99): ref_feat (R, D) float32."""
    ref_feat = np.asarray(ref_feat, dtype=np.float32)
    mean = np.mean(ref_feat, axis=0)
    thres = 0
    for i in range(ref_feat.shape[0]):
        thres = max(thres, np.linalg.norm(mean - ref_feat[i]))
    return mean, np.float32(thres)


def l2_filter(E, mean, thres):
    """filter loop (filter_faces_using_reference.py:186-189): keep iff ||e - mean|| <= thres."""
    E = np.asarray(E, dtype=np.float32)
    dist = np.array([np.linalg.norm(e - mean) for e in E], dtype=np.float32)
    return dist, dist <= thres


def cosine_distance(a, b):
    """fde/face_extraction/extract_and_label_faces_from_dataset.py:106: 1 - <a,b>/(|a||b|)."""
    return 1 - np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))


def cosine_filter(G, R, tau):
    """SURVEY S4 (build-defined generalisation of the cosine above): per gallery row the best cosine
    similarity over the reference rows, its argmax (lowest index on ties) and keep = best >= tau."""
    G, R = torch.as_tensor(G, dtype=torch.float32), torch.as_tensor(R, dtype=torch.float32)
    S = (G @ R.T) / (G.norm(dim=1, keepdim=True) * R.norm(dim=1, keepdim=True).T)
    best, arg = S.max(dim=1)
    return best.numpy(), arg.numpy().astype(np.int32), (best >= tau).numpy(), S.numpy()
