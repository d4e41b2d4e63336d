"""Shared input generators for the parity tests (restated from the reference's unit tests)."""
import numpy as np


def cos_abs(a, b):
    """|cos| between rows of two (N,3) arrays."""
    num = np.abs(np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1))
    den = np.linalg.norm(a.astype(np.float64), axis=1) * np.linalg.norm(b.astype(np.float64), axis=1)
    return num / np.maximum(den, 1e-300)


def frob(Ta, Tb, to_matrix):
    return float(np.linalg.norm(to_matrix(Ta).astype(np.float64) - to_matrix(Tb).astype(np.float64)))


def sphere_cloud(n):
    """make_sphere_cloud (registration.rs:1148-1165): Fibonacci sphere r=3, normals = unit position."""
    radius = np.float32(3.0)
    golden = np.float32(np.pi) * (np.float32(3.0) - np.sqrt(np.float32(5.0)))
    pts, nrm = [], []
    for i in range(n):
        y = np.float32(1.0) - (np.float32(i) / max(np.float32(n) - np.float32(1.0), np.float32(1.0))) * np.float32(2.0)
        r = np.sqrt(max(np.float32(1.0) - y * y, np.float32(0.0)))
        th = golden * np.float32(i)
        x, z = np.cos(th) * r, np.sin(th) * r
        nrm.append([x, y, z])
        pts.append([x * radius, y * radius, z * radius])
    return np.array(pts, np.float32), np.array(nrm, np.float32)


def quat_z(angle):
    h = np.float32(angle) / np.float32(2)
    return np.array([0, 0, np.sin(h), np.cos(h)], np.float32)


def rotate_z(points, angle):
    c, s = np.cos(np.float32(angle)), np.sin(np.float32(angle))
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    return (points @ R.T).astype(np.float32)
