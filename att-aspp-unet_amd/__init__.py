"""MI355X-native Attention-ASPP-UNet hot path."""
