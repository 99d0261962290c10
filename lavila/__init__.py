"""Drop-in import path: `lavila.models.*` resolves to the MI355X-native implementation in `lavila_amd`."""
