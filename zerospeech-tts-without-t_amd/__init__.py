"""zs_amd: MI355X-native ASR-TTS autoencoder hot path."""
