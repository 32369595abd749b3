"""MI355X-native flexibility-provision environment + safe-MADDPG hot path."""
