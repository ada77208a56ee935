"""merlin_amd: MI355X-native hot path for Merlin (MMGPT) fwd/bwd."""
