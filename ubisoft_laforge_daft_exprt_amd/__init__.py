"""MI355X-native forward/backward path for the Daft-Exprt acoustic model."""
