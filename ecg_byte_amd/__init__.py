"""ecg-byte_amd: MI355X-native hot path of ECG-Byte (quantise -> BPE encode -> assemble)."""
