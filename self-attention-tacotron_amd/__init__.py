"""MI355X-native teacher-forced training path of Self-attention Tacotron."""
