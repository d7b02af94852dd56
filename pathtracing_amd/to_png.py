"""Accumulation buffers -> PNG, the display transform of the reference's python_utils/to_png.py:
tone-map c / (max_c + 1) (:18-19), clip to [0, 1] (:37), then linear -> sRGB iff the settings ask for
gamma correction (:53-56, default False when the key is absent, :74).

    python -m pathtracing_amd.to_png --name result.png --scene_directory scenes/example [--temp temp] [--out images]
"""
import argparse
import json
import os

import numpy as np


def tone_map(image):
    return image / (np.max(image, axis=-1) + 1)[:, :, None]


def gamma_correction(image):
    threshold = 0.0031308
    return np.where(image <= threshold, image * 12.92, 1.055 * np.power(np.maximum(image, 0), 1.0 / 2.4) - 0.055)


def load_image_data(file_name, width, height):
    image = np.fromfile(file_name, dtype=np.float64).reshape(height, width, 3)
    if np.isnan(image).any():
        print(f"Found NaN pixels in file: {file_name}")
    return np.clip(tone_map(image), 0, 1)


def display_image(raw, use_gamma_correction):
    """float64 [H, W, 3] radiance -> float [H, W, 3] in [0, 1] ready to save."""
    image = np.clip(tone_map(np.asarray(raw, dtype=np.float64)), 0, 1)
    return gamma_correction(image) if use_gamma_correction else image


def denoising_enabled(denoising_settings):
    return any(task.get("mode", "skip") != "skip" for task in denoising_settings.get("pipeline", []))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", required=True)
    ap.add_argument("--scene_directory", required=True)
    ap.add_argument("--temp", default="temp")
    ap.add_argument("--out", default="images")
    args = ap.parse_args()
    import matplotlib.pyplot as plt
    with open(os.path.join(args.scene_directory, "settings.json")) as f:
        settings = json.load(f)
    width, height = settings.get("WIDTH", 1000), settings.get("HEIGHT", 1000)
    gamma = settings.get("use_gamma_correction", False)
    os.makedirs(args.out, exist_ok=True)
    image = load_image_data(os.path.join(args.temp, "raw_pixel.dat"), width, height)
    plt.imsave(os.path.join(args.out, args.name), gamma_correction(image) if gamma else image)
    with open(os.path.join(args.scene_directory, "denoising.json")) as f:
        denoising = json.load(f)
    if denoising_enabled(denoising):
        os.makedirs(os.path.join(args.out, "denoised"), exist_ok=True)
        image = load_image_data(os.path.join(args.temp, "raw_denoised.dat"), width, height)
        plt.imsave(os.path.join(args.out, "denoised", args.name), gamma_correction(image) if gamma else image)


if __name__ == "__main__":
    main()
