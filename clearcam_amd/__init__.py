"""clearcam_amd — MI355X-native detect / CLIP-encode / search path for clearcam."""

