"""Drop-in alias: ``from dICP.ICP import ICP`` resolves to the MI355X-native dicp_amd package."""
