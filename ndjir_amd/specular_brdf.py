"""Specular BRDF models (white light).

Reference: python/specular_brdf.py -- dot :23-37, filament_specular_brdf :40-118,
ue4_specular_brdf :121-191, specular_brdf_model :194-199.
"""
import math

import torch


def dot(u, v, with_mask=False, eps=1e-8):
    """specular_brdf.py:23-37: clamped dot product (+ mask of un-clamped entries, no gradient)."""
    uv = (u * v).sum(-1, keepdim=True)
    mask = (uv > eps).to(uv.dtype).detach()
    uv = uv.clamp(min=eps)
    return (uv, mask) if with_mask else uv


def _bcast(normal, view_dir, light_dir, roughness, specular_color):
    B, R, _ = normal.shape
    M = light_dir.shape[2]
    return (normal.reshape(B, R, 1, 3).expand(B, R, M, 3), view_dir.reshape(B, R, 1, 3).expand(B, R, M, 3),
            roughness.reshape(B, R, 1, 1).expand(B, R, M, 1),
            specular_color.reshape(B, R, 1, -1).expand(B, R, M, specular_color.shape[-1]))


def _half(light_dir, view_dir):
    h = light_dir + view_dir
    return h / torch.sqrt((h * h).sum(-1, keepdim=True))


def filament_specular_brdf(normal, view_dir, light_dir, roughness, specular_color, conf):
    """specular_brdf.py:40-118.  All direction vectors unit length.
    normal (B,R,3), view_dir (B,R,1,3), light_dir (B,R,M,3), roughness (B,R,1), specular_color (B,R,3)."""
    normal, view_dir, roughness, specular_color = _bcast(normal, view_dir, light_dir, roughness, specular_color)
    half_dir = _half(light_dir, view_dir)
    a2 = roughness ** 2
    eps_dot = conf.renderer.eps_dot
    nol, m_nol = dot(normal, light_dir, True, eps_dot)
    nov, m_nov = dot(normal, view_dir, True, eps_dot)
    noh, m_noh = dot(normal, half_dir, True, eps_dot)
    voh = dot(view_dir, half_dir, False, eps_dot)
    eps = 1e-6

    def V1(nou):
        return 1 / (nou + (a2 + (1 - a2) * nou ** 2) ** 0.5 + eps)

    V = V1(nol) * V1(nov)
    Fs = specular_color + (1 - specular_color) * (1 - voh) ** 5
    if conf.specular_brdf.sampling == "importance":
        sBRDF = V * Fs * (4 * voh / noh)
    else:
        D = a2 / (math.pi * (noh ** 2 * (a2 - 1) + 1) ** 2 + eps)
        sBRDF = math.pi * D * V * Fs
    return sBRDF * (m_nol * m_nov * m_noh), nol


def ue4_specular_brdf(normal, view_dir, light_dir, roughness, specular_color, conf):
    """specular_brdf.py:121-191."""
    normal, view_dir, roughness, specular_color = _bcast(normal, view_dir, light_dir, roughness, specular_color)
    half_dir = _half(light_dir, view_dir)
    a = roughness ** 2
    a2 = a ** 2
    eps_dot = conf.renderer.eps_dot
    nol, m_nol = dot(normal, light_dir, True, eps_dot)
    nov, m_nov = dot(normal, view_dir, True, eps_dot)
    noh, m_noh = dot(normal, half_dir, True, eps_dot)
    voh = dot(view_dir, half_dir, False, eps_dot)
    eps = 1e-6
    k = (roughness + 1) ** 2 / 8

    def G1(nou):
        return nou / (nou * (1 - k) + k + eps)

    G = G1(nol) * G1(nov)
    Fs = specular_color + (1 - specular_color) * 2 ** ((-5.55473 * voh - 6.98316) * voh)
    if conf.specular_brdf.sampling == "importance":
        sBRDF = G * Fs * (voh / (noh * nov))
    else:
        D = a2 / (math.pi * (noh ** 2 * (a2 - 1) + 1) ** 2 + eps)
        sBRDF = math.pi * D * G * Fs / (4 * nov * nol)
    return sBRDF * (m_nol * m_nov * m_noh), nol


def specular_brdf_model(normal, view_dir, light_dir, roughness, specular_color, conf):
    models = dict(filament=filament_specular_brdf, ue4=ue4_specular_brdf)
    return models[conf.specular_brdf.model](normal, view_dir, light_dir, roughness, specular_color, conf)
