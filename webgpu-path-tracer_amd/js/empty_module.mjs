export default {};
